#!/bin/bash
# Build libtitok_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")"
OUT=../libtitok_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wall -Wno-unused-function"
mkdir -p build
pids=()
for f in ttv_elem ttv_gemm ttv_attn ttv_attn_swp ttv_attn64 ttv_mlp ttv_bwd ttv_train ttv_vq ttv_api; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ ttv_common.h -nt build/$f.o ] || [ ttv_kernels.h -nt build/$f.o ] || { [ $f = ttv_gemm ] && { [ ttv_qkv256.inc -nt build/$f.o ] || [ ttv_qkv256ws.inc -nt build/$f.o ]; }; } || [ ../../include/titok_hip.h -nt build/$f.o ]; then
    if [ $f = ttv_attn64 ]; then bash build_attn64.sh build/$f.o &       # two-step build: see build_attn64.sh
    else hipcc $FLAGS -c $f.hip -o build/$f.o &
    fi
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p || exit 1; done
hipcc --offload-arch=gfx950 -shared -fPIC build/ttv_elem.o build/ttv_gemm.o build/ttv_attn.o build/ttv_attn_swp.o build/ttv_attn64.o build/ttv_mlp.o build/ttv_bwd.o build/ttv_train.o build/ttv_vq.o build/ttv_api.o -o $OUT
echo "built $(realpath $OUT)"
