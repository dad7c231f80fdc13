#!/bin/bash
# GEMM k-step variants, each in its own build directory / output (the default build is untouched)
R=${GRAFT_REPO_ROOT:-/root/repo}
for bk in 16 32; do
  EXTRA_FLAGS="-DBK=$bk" BUILD_DIR=/tmp/nhp_build_bk$bk NHP_LIB_OUT=/tmp/libnhp_bk$bk.so bash $R/networkhawkesprocesses.jl_amd/csrc/build.sh > /dev/null 2>&1
  echo "BK=$bk"
  (cd $R && export NHP_LIB=/tmp/libnhp_bk$bk.so && python -m pytest tests/test_discrete_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -1; cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bk/$bk -- python3 $R/tools/c4.py > /dev/null 2>&1; grep k_gemm $R/gpurun_out/prof_bk/$bk/*/*kernel_stats.csv | cut -d'"' -f2,3 | cut -c1-80)
done
