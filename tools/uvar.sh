#!/bin/bash
# build variants with different in-flight depth U (separate object dirs / outputs: the default build is untouched) and time them
R=${GRAFT_REPO_ROOT:-/root/repo}
for us in 2 4 8; do for um in 2 4; do
  EXTRA_FLAGS="-DNHP_U_SMALL=$us -DNHP_U_MID=$um" BUILD_DIR=/tmp/nhp_build_u${us}_${um} NHP_LIB_OUT=/tmp/libnhp_u${us}_${um}.so \
    bash $R/networkhawkesprocesses.jl_amd/csrc/build.sh > /dev/null 2>&1
  echo "U_SMALL=$us U_MID=$um"
  (cd $R && export NHP_LIB=/tmp/libnhp_u${us}_${um}.so && python tools/kbench.py windowed_k8 30 && python tools/kbench.py windowed_k64 20 && python tools/kbench.py windowed_k512 8)
done; done
