#!/bin/bash
# kernel time of k_disc_convolve under variant libraries: gpurun -- 'bash tools/dbg/convk.sh <tag> ...'
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  L=$R/networkhawkesprocesses.jl_amd/libnhp_v$v.so; [ "$v" = base ] && L=$R/networkhawkesprocesses.jl_amd/libnhp.so
  NHP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/convk_$v -- python3 $R/tools/convbench.py > /dev/null 2>&1
  python3 - "$v" <<PY
import csv,glob,sys
for f in glob.glob('$R/gpurun_out/convk_%s/*/*kernel_stats.csv'%sys.argv[1]):
    for r in csv.DictReader(open(f)):
        if 'k_disc_convolve' in r['Name']: print('variant %-8s %-30s calls %s avg %.1f us' % (sys.argv[1], r['Name'][:30], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
