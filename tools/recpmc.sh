#!/bin/bash
# PMC passes over the full recursion (k_recursive*): where do the cycles go?  Usage: tools/recpmc.sh  (on the GPU box)
# (Round 2's run of this script left gpurun_out/pmc_rec/s1-s3 only: the fourth set below produced no counter file on
#  k_recursive_waves and no log of that pass was kept -- stdout / stderr go to /dev/null here -- so its cause is not known;
#  profiles/r02_pmc/recursive_waves_counters.txt holds the three sets that finished.  Not re-run.)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_rec/s$i -- python3 $R/tools/kbench.py recursive_full 3 > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob('$R/gpurun_out/pmc_rec/s$i/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_recursive' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(f"{k:28s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
done
