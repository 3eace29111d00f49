#!/bin/bash
# Build libmi355rec.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [-j N]
set -e
cd "$(dirname "$0")"
ROOT=../..
OUT=libmi355rec.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I${ROOT}/include -Wno-unused-result"
mkdir -p obj
pids=()
for f in *.hip; do
  o=obj/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ ${ROOT}/include/mi355rec.h -nt "$o" ]; then
    hipcc $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC obj/*.o -o $OUT
echo "built $(pwd)/$OUT"
