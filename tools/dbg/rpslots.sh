#!/bin/bash
# k_disc_resample_parents by slots per thread (NHP_RP_SLOTS) and data rate: gpurun -- 'bash tools/dbg/rpslots.sh'
for sl in 4 2 1; do for r in 0.05 0.1; do echo "slots $sl rate $r"; NHP_RP_SLOTS=$sl DG_RATE=$r bash tools/kstats.sh rp_${sl}_$r tools/dgibbs.py | grep k_disc_resample; done; done
