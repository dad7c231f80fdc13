#!/bin/bash
for c in 512 1024 2048; do for g in 4 8; do NHP_CHUNK=$c NHP_GROUP=$g python tools/kbench.py windowed_k8 30; done; done
for g in 8 16; do NHP_GROUP=$g python tools/kbench.py windowed_k64 20; done
for g in 16 32; do NHP_GROUP=$g python tools/kbench.py windowed_k512 10; done
NHP_DBG=1 python tools/kbench.py windowed_k8 30
NHP_DBG=8 python tools/kbench.py windowed_k8 30
python tools/kbench.py logitnormal_k8 20
