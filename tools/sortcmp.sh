#!/bin/bash
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_cont_loglik_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -1
for m in 0 1 2; do
  NHP_SORT=$m python tools/kbench.py windowed_k8 30
  NHP_SORT=$m python tools/kbench.py windowed_k64 20
done
cd /tmp; export TMPDIR=/tmp
for m in 0 1 2; do
  NHP_SORT=$m rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_sort/m$m -- python3 $R/tools/kbench.py windowed_k8 5 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
v=[float(r['Counter_Value']) for f in glob.glob('$R/gpurun_out/pmc_sort/m$m/*/*counter_collection.csv') for r in csv.DictReader(open(f)) if 'k_windowed' in r['Kernel_Name']]
print('NHP_SORT=$m FETCH_SIZE KB mean', sum(v)/len(v))
PY
done
