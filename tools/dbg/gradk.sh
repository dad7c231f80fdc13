#!/bin/bash
# per-kernel times of (log-likelihood, gradient) at the metric size: gpurun -- 'bash tools/dbg/gradk.sh <outdir>'
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-gradk}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/tools/gradbench.py > $OUT/gradbench.log 2>&1
cat $OUT/gradbench.log | grep -v amdgpu.ids
f=$(ls $OUT/prof/*/*kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
