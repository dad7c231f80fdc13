#!/bin/bash
# Timing ablations and stamps of the discrete adjacency step kernel (k_dadj_step).  Run on the GPU box:
#   gpurun -- 'bash tools/dbg/dadjabl.sh'          (the variant libraries are built here first: bash tools/dbg/dadjabl.sh build)
# DADJ_ABL: 1 no entry loop, 2 no log ratio, 6 two logarithms instead of the series, 7 no flip carry; -DDADJ_STAMP: s_memrealtime
# stamps of one step (NHP_DADJ_STAMP_STEP, default 100) -> tools/dbg/dadjstamps.py
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then
  cd $R/networkhawkesprocesses.jl_amd/csrc
  for v in 1 2 6 7; do EXTRA_FLAGS="-DDADJ_ABL=$v" BUILD_DIR=/tmp/build_abl$v NHP_LIB_OUT=$R/gpurun_in_abl$v.so bash build.sh | tail -1; done
  EXTRA_FLAGS="-DDADJ_STAMP" BUILD_DIR=/tmp/build_stamp NHP_LIB_OUT=$R/gpurun_in_stamp.so bash build.sh | tail -1
  exit 0
fi
cd $R
for v in 1 2 6 7; do echo "abl=$v"; NHP_LIB=$R/gpurun_in_abl$v.so DG_RATE=0.05 timeout -k 10 120 python tools/dadj.py 2>&1 | grep -v amdgpu | tail -1; done
echo full; DG_RATE=0.05 python tools/dadj.py 2>&1 | grep -v amdgpu | tail -1
NHP_DADJ_STAMPS=/tmp/st.bin NHP_LIB=$R/gpurun_in_stamp.so DG_RATE=0.05 python tools/dadj.py 2>&1 | grep -v amdgpu | tail -1
python tools/dbg/dadjstamps.py /tmp/st.bin
