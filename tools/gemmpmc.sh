#!/bin/bash
# MFMA-pipe counters of the fp64 GEMM kernels: gpurun -- 'bash tools/gemmpmc.sh'  (tools/c4.py under rocprofv3 --pmc)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/gemmpmc/s$i -- python3 $R/tools/c4.py > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$R/gpurun_out/gemmpmc/s*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_gemm' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    print(k)
    for c,v in sorted(acc[k].items()): print(f"   {c:28s} {sum(v)/len(v):.5g}")
PY
