#!/bin/bash
# per-kernel time of a python tool: tools/kstats.sh <outdir-name> <script> [args]   (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
S=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$OUT -- python3 $R/$S "$@" > $R/gpurun_out/$OUT.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('$R/gpurun_out/$OUT/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s}  avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  {r['Percentage']}%")
PY
tail -3 $R/gpurun_out/$OUT.log
