#!/bin/bash
# Builds libnhp.so (gfx950) next to the package.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libnhp.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wall -Wno-unused-function"
mkdir -p build
pids=()
for f in *.hip; do
  o=build/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ nhp_internal.h -nt "$o" ] || [ nhp_math.h -nt "$o" ] || [ ../../include/nhp.h -nt "$o" ]; then
    hipcc $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -fPIC -shared -o "$OUT" build/*.o
echo "built $(realpath $OUT)"
