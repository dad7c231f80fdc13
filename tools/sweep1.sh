#!/bin/bash
# group-width / chunk sweep of the windowed kernel
for wl in windowed_k8 windowed_k64 windowed_k512; do
  for g in 1 2 4 8 16 32 64; do
    NHP_GROUP=$g python tools/kbench.py $wl 20
  done
done
for c in 64 128 256 512 1024; do NHP_CHUNK=$c python tools/kbench.py windowed_k8 20; done
python tools/kbench.py recursive 5
python tools/kbench.py logitnormal_k8 20
