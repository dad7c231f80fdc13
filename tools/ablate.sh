#!/bin/bash
# needs a build with -DNHP_ABLATE (add it to FLAGS in csrc/build.sh)
for d in 0 1 2 3 4 7 8 15; do echo "NHP_DBG=$d"; NHP_DBG=$d python tools/kbench.py windowed_k8 30; done
