#!/bin/bash
# PMC passes over one kernel: tools/pmc.sh <kernel-name-substring> <outdir-name> -- <python script and args>
# (each counter set in its own rocprofv3 run; --kernel-trace only, as the pool requires)
R=${GRAFT_REPO_ROOT:-/root/repo}
KERN=$1; OUT=$2; shift 3
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/$OUT/s$i -- python3 "$@" > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob('$R/gpurun_out/$OUT/s$i/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if '$KERN' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(f"{k:32s} {sum(v)/len(v):.5g}  (n={len(v)})")
PY
done
