#!/bin/bash
# time-partitioned items (NHP_XCD) x child ordering (NHP_SORT): time per evaluation and FETCH_SIZE of k_windowed at K = 8
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for x in 0 2 4; do for so in 2 1 0; do
  export NHP_XCD=$x NHP_SORT=$so
  t=$(python3 $R/tools/kbench.py windowed_k8 30 2>/dev/null | awk '{print $5}')
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_xs/x${x}s${so} -- python3 $R/tools/kbench.py windowed_k8 5 > /dev/null 2>&1
  f=$(python3 - <<PY
import csv,glob
v=[float(r['Counter_Value']) for f in glob.glob('$R/gpurun_out/pmc_xs/x${x}s${so}/*/*counter_collection.csv') for r in csv.DictReader(open(f)) if 'k_windowed' in r['Kernel_Name']]
print(round(2*sum(v)/len(v)/1024,1) if v else 'na')
PY
)
  echo "NHP_XCD=$x NHP_SORT=$so  $t us/eval  traffic ${f} MB"
done; done
