#!/bin/bash
# build variants with different in-flight depth U and time them
cd "$GRAFT_REPO_ROOT/networkhawkesprocesses.jl_amd/csrc"
for us in 2 4 8; do for um in 2 4; do
  rm -rf build; sed -i "s/^FLAGS=.*/FLAGS=\"--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wall -Wno-unused-function -DNHP_U_SMALL=$us -DNHP_U_MID=$um\"/" build.sh
  ./build.sh > /dev/null 2>&1
  echo "U_SMALL=$us U_MID=$um"
  (cd $GRAFT_REPO_ROOT && python tools/kbench.py windowed_k8 30 && python tools/kbench.py windowed_k64 20 && python tools/kbench.py windowed_k512 8)
done; done
