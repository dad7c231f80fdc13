#!/bin/bash
# k_windowed_slices by (workgroup size, rows per request): gpurun -- 'bash tools/dbg/slicesweep.sh [workload]'
W=${1:-windowed_k8}
echo -n "pairs (NHP_SLICES=0)  "; NHP_SLICES=0 python tools/kbench.py $W 100 | awk '{print $5, $6, $NF}'
for b in 64 128 256 512 1024; do for c in 2 4; do
  echo -n "BLOCK=$b C=$c  "; NHP_SLICES_CFG=$b,$c python tools/kbench.py $W 100 | awk '{print $5, $6, $NF}'
done; done
