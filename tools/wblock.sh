#!/bin/bash
cd "$GRAFT_REPO_ROOT/networkhawkesprocesses.jl_amd/csrc"
for wb in 256 512 1024; do for us in 2 4; do
  rm -rf build; sed -i "s/^FLAGS=.*/FLAGS=\"--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wall -Wno-unused-function -DNHP_WBLOCK=$wb -DNHP_U_SMALL=$us -DNHP_U_MID=$us\"/" build.sh
  ./build.sh > /dev/null 2>&1
  echo "WBLOCK=$wb U=$us"
  (cd $GRAFT_REPO_ROOT && python -m pytest tests/test_cont_loglik_gpu.py -m gpu -x -q 2>&1 | tail -1; python tools/kbench.py windowed_k8 30 && python tools/kbench.py windowed_k64 20 && python tools/kbench.py windowed_k512 8)
done; done
