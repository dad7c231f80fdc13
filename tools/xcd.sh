#!/bin/bash
R=$GRAFT_REPO_ROOT
for x in 0 2 4; do for m in 1 2; do
  echo "NHP_XCD=$x NHP_SORT=$m"
  NHP_XCD=$x NHP_SORT=$m python tools/kbench.py windowed_k8 30
  NHP_XCD=$x NHP_SORT=$m python tools/kbench.py windowed_k64 20
done; done
cd /tmp; export TMPDIR=/tmp
for x in 0 2 4; do
  NHP_XCD=$x NHP_SORT=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_xcd/x$x -- python3 $R/tools/kbench.py windowed_k8 5 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
v=[float(r['Counter_Value']) for f in glob.glob('$R/gpurun_out/pmc_xcd/x$x/*/*counter_collection.csv') for r in csv.DictReader(open(f)) if 'k_windowed' in r['Kernel_Name']]
print('NHP_XCD=$x SORT=1 FETCH_SIZE KB mean', sum(v)/len(v))
PY
done
