#!/bin/bash
# discrete Gibbs parent counts: node tiles of one bin range on one XCD (NHP_RP_XCD=1) x checkpointed second walk (NHP_RP_CHK=1), kernel time from rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do for ck in 0 1; do for cfg in "128,128,512 2" "64,128,256 2" "128,256,1024 2"; do
  set -- $cfg
  d=$R/gpurun_out/rpxcd/${x}_${ck}_${1//,/_}_$2
  NHP_RP_XCD=$x NHP_RP_CHK=$ck NHP_RP_TILE=$1 NHP_RP_SLOTS=$2 DG_RATE=${DG_RATE:-0.05} rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/dgibbs.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("$d/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "resample_parents" in r["Name"]:
            print("xcd $x chk $ck tile $1 slots $2: avg %.2f ms" % (float(r["AverageNs"]) / 1e6))
PY
done; done; done
