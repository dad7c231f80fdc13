#!/bin/bash
# Compile-time ablations of k_adj_sweep (results are wrong by design; timing only).  Build the variants first:
#   for v in 0 1 2 4 7; do EXTRA_FLAGS=-DADJ_ABL=$v BUILD_DIR=$PWD/networkhawkesprocesses.jl_amd/csrc/build_abl$v \
#       NHP_LIB_OUT=$PWD/networkhawkesprocesses.jl_amd/libnhp_abl$v.so bash networkhawkesprocesses.jl_amd/csrc/build.sh; done
# bit 1: no global requests in the loop, 2: no LDS read of the child's intensity, 4: no store of the decision
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1 2 4 7; do
  NHP_LIB=$R/networkhawkesprocesses.jl_amd/libnhp_abl$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/adjabl$v -- python3 $R/tools/adj.py > /dev/null 2>&1
done
python3 - <<PY
import csv,glob
for d in (0,1,2,4,7):
    for f in glob.glob('$R/gpurun_out/adjabl%d/*/*kernel_stats.csv'%d):
        for r in csv.DictReader(open(f)):
            if 'k_adj_sweep' in r['Name']: print('ADJ_ABL=%d'%d, r['Calls'], r['AverageNs'], r['MinNs'])
PY
