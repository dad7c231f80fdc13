#!/bin/bash
# discrete Gibbs parent counts by tile (bins x nodes, threads) and slots per thread: kernel time from rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tile in 64,128,256 128,128,512 128,256,1024; do for sl in 1 2 4; do
  d=$R/gpurun_out/rptile/${tile//,/_}_$sl
  NHP_RP_TILE=$tile NHP_RP_SLOTS=$sl DG_RATE=0.05 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/dgibbs.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("$d/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "resample_parents" in r["Name"]:
            print("tile $tile slots $sl:", r["Name"][:60], "avg %.2f ms" % (float(r["AverageNs"]) / 1e6))
PY
done; done
