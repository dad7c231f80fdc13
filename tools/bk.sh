#!/bin/bash
cd "$GRAFT_REPO_ROOT/networkhawkesprocesses.jl_amd/csrc"
for bk in 16 32; do
  rm -f build/disc.o; sed -i "s/^FLAGS=.*/FLAGS=\"--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wall -Wno-unused-function -DBK=$bk\"/" build.sh
  ./build.sh > /dev/null 2>&1
  echo "BK=$bk"
  (cd $GRAFT_REPO_ROOT && python -m pytest tests/test_discrete_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -1; cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bk/$bk -- python3 $GRAFT_REPO_ROOT/tools/c4.py > /dev/null 2>&1; grep k_gemm $GRAFT_REPO_ROOT/gpurun_out/prof_bk/$bk/*/*kernel_stats.csv | cut -d'"' -f2,3 | cut -c1-80)
done
