#!/bin/bash
for wl in windowed_k8 windowed_k64 windowed_k512; do
  for g in 2 4 8 16 32; do
    NHP_GROUP=$g python tools/kbench.py $wl 20
  done
done
python tools/kbench.py windowed_k8 50
python tools/kbench.py logitnormal_k8 20
