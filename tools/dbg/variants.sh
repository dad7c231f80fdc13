#!/bin/bash
# Per-kernel times of a tool script under variant builds of the library (libnhp_v<tag>.so, see tools/README.md):
#   gpurun -- 'bash tools/dbg/variants.sh <kernel-substring> <script> <tag> [<tag> ...]'
R=$GRAFT_REPO_ROOT
K=$1; S=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  NHP_LIB=$R/networkhawkesprocesses.jl_amd/libnhp_v$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/var_$v -- python3 $R/$S > /dev/null 2>&1
done
python3 - "$K" "$@" <<PY
import csv,glob,sys
for d in sys.argv[2:]:
    for f in glob.glob('$R/gpurun_out/var_%s/*/*kernel_stats.csv'%d):
        for r in csv.DictReader(open(f)):
            if sys.argv[1] in r['Name']: print('variant %-6s %-44s calls %s avg %.1f us' % (d, r['Name'][:44], r['Calls'], float(r['AverageNs'])/1e3))
PY
