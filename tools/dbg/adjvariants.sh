#!/bin/bash
# Time k_adj_sweep in variant builds (tools/adj.py under rocprofv3 --kernel-trace --stats).
#   build:  for v in 3 4 6; do EXTRA_FLAGS=-DADJ_DEPTH=$v BUILD_DIR=$PWD/networkhawkesprocesses.jl_amd/csrc/build_v$v \
#               NHP_LIB_OUT=$PWD/networkhawkesprocesses.jl_amd/libnhp_v$v.so bash networkhawkesprocesses.jl_amd/csrc/build.sh; done
#   run  :  gpurun -- 'bash tools/dbg/adjvariants.sh 3 4 6'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  NHP_LIB=$R/networkhawkesprocesses.jl_amd/libnhp_v$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/adjv$v -- python3 $R/tools/adj.py > /dev/null 2>&1
done
python3 - "$@" <<PY
import csv,glob,sys
for d in sys.argv[1:]:
    for f in glob.glob('$R/gpurun_out/adjv%s/*/*kernel_stats.csv'%d):
        for r in csv.DictReader(open(f)):
            if 'k_adj_sweep' in r['Name']: print('variant %s'%d, r['Calls'], r['AverageNs'], r['MinNs'])
PY
