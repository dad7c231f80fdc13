#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for d in 0 8 1 3 15; do
  NHP_DBG=$d rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_abl/d$d -- python3 $R/tools/kbench.py windowed_k8 5 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
v=[float(r['Counter_Value']) for f in glob.glob('$R/gpurun_out/pmc_abl/d$d/*/*counter_collection.csv') for r in csv.DictReader(open(f)) if 'k_windowed' in r['Kernel_Name']]
print('NHP_DBG=$d FETCH_SIZE KB mean', sum(v)/len(v))
PY
done
