#!/bin/bash
# needs an ablation build: EXTRA_FLAGS=-DNHP_ABLATE BUILD_DIR=/tmp/nhp_build_abl NHP_LIB_OUT=/tmp/libnhp_abl.so bash networkhawkesprocesses.jl_amd/csrc/build.sh; export NHP_LIB=/tmp/libnhp_abl.so
for d in 0 1 2 3 4 7 8 15; do echo "NHP_DBG=$d"; NHP_DBG=$d python tools/kbench.py windowed_k8 30; done
