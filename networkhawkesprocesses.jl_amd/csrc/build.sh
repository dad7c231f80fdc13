#!/bin/bash
# Builds libnhp.so (gfx950) next to the package.  hipcc cross-compiles without a GPU.
#   EXTRA_FLAGS="-DNHP_WBLOCK=512" BUILD_DIR=build_wb512 NHP_LIB_OUT=/tmp/libnhp_wb512.so ./build.sh
# builds a variant into its own object directory and output file (load it with NHP_LIB=<path>); the default build is
# never edited by the sweep scripts in tools/.
set -e
cd "$(dirname "$0")"
OUT=${NHP_LIB_OUT:-../libnhp.so}
BUILD=${BUILD_DIR:-build}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wall -Wno-unused-function ${EXTRA_FLAGS}"
mkdir -p "$BUILD"
# objects are stale when the flags changed, not only when a source or header is newer
sig=$(echo "$FLAGS" | md5sum | cut -c1-16)
if [ "$(cat "$BUILD/.flags" 2>/dev/null)" != "$sig" ]; then rm -f "$BUILD"/*.o; echo "$sig" > "$BUILD/.flags"; fi
pids=()
for f in *.hip; do
  o=$BUILD/${f%.hip}.o
  stale=0
  [ -f "$o" ] || stale=1
  for dep in "$f" nhp_internal.h nhp_math.h nhp_rng.h nhp_lbfgs.h ../../include/nhp.h; do [ "$dep" -nt "$o" ] && stale=1; done
  if [ $stale = 1 ]; then
    hipcc $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -fPIC -shared -o "$OUT" "$BUILD"/*.o -ldl
echo "built $(realpath $OUT)"
