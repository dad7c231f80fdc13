#!/bin/bash
# workgroup size x in-flight depth variants of the windowed kernel, each in its own build directory / output
R=${GRAFT_REPO_ROOT:-/root/repo}
for wb in 256 512 1024; do for us in 2 4; do
  EXTRA_FLAGS="-DNHP_WBLOCK=$wb -DNHP_U_SMALL=$us -DNHP_U_MID=$us" BUILD_DIR=/tmp/nhp_build_wb${wb}_${us} NHP_LIB_OUT=/tmp/libnhp_wb${wb}_${us}.so \
    bash $R/networkhawkesprocesses.jl_amd/csrc/build.sh > /dev/null 2>&1
  echo "WBLOCK=$wb U=$us"
  (cd $R && export NHP_LIB=/tmp/libnhp_wb${wb}_${us}.so && python -m pytest tests/test_cont_loglik_gpu.py -m gpu -x -q 2>&1 | tail -1; python tools/kbench.py windowed_k8 30 && python tools/kbench.py windowed_k64 20 && python tools/kbench.py windowed_k512 8)
done; done
