#!/bin/bash
# k_windowed_slices by (workgroup size, rows per request, masked rows): gpurun -- 'bash tools/dbg/slicesweep.sh [workload]'
W=${1:-windowed_k8}
echo -n "pairs (NHP_SLICES=0)  "; NHP_SLICES=0 python tools/kbench.py $W 100 2>/dev/null | awk '{print $5, $6, $NF}'
for b in ${BLOCKS:-256 512 1024}; do for c in 2 4; do for m in 0 1; do
  echo -n "BLOCK=$b C=$c MASKED=$m  "; NHP_SLICES_CFG=$b,$c,$m python tools/kbench.py $W 100 2>/dev/null | awk '{print $5, $6, $NF}'
done; done; done
