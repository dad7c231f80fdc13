#!/bin/bash
# k_windowed_pairs by (lanes per child, children per group in flight, workgroup size): gpurun -- 'bash tools/dbg/pairsweep.sh'
for b in 256 512 1024; do for g in 2 4 8; do for u in 1 2 4; do
  [ $b = 1024 ] && [ $u = 4 ] && continue
  echo -n "G=$g U=$u BLOCK=$b  "; NHP_PAIRS_CFG=$g,$u,$b python tools/kbench.py ${1:-windowed_k8} 100 | awk '{print $5, $6}'
done; done; done
