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

# TORCH_LIBRARY(mi355rec) registration of the hot operators over the same C ABI: host-only C++ (g++), linked against the
# PyTorch of this image and libmi355rec.so next to it
TORCH_DIR=$(python3 -c "import os, torch; print(os.path.dirname(torch.__file__))")
TOUT=libmi355rec_torch.so
if [ ! -f "$TOUT" ] || [ torch_ops.cpp -nt "$TOUT" ] || [ ${ROOT}/include/mi355rec.h -nt "$TOUT" ] || [ "$OUT" -nt "$TOUT" ]; then
  g++ -O2 -fPIC -shared -std=c++17 -D__HIP_PLATFORM_AMD__ -DUSE_ROCM -D_GLIBCXX_USE_CXX11_ABI=1 \
      -I${ROOT}/include -I${TORCH_DIR}/include -I${TORCH_DIR}/include/torch/csrc/api/include -I/opt/rocm/include \
      torch_ops.cpp -o $TOUT -L${TORCH_DIR}/lib -ltorch -ltorch_cpu -lc10 -lc10_hip -L. -l:libmi355rec.so \
      -Wl,-rpath,'$ORIGIN' -Wl,-rpath,${TORCH_DIR}/lib
fi
echo "built $(pwd)/$TOUT"
