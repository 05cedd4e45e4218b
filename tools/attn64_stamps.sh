#!/bin/bash
# Diagnostic build of the 64-rows-per-wave attention kernel with -DATTN64_STAMPS (extra defines: $1) linked against the product objects
# -> build/libtitok_hip_stamps64.so
set -e
cd "$(dirname "$0")/../titok_video_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wall -Wno-unused-function"
bash build_attn64.sh build/ttv_attn64_stamps.o -DATTN64_STAMPS $1
hipcc --offload-arch=gfx950 -shared -fPIC build/ttv_elem.o build/ttv_gemm.o build/ttv_attn.o build/ttv_attn64_stamps.o build/ttv_mlp.o build/ttv_bwd.o build/ttv_train.o build/ttv_vq.o build/ttv_api.o -o build/libtitok_hip_stamps64.so
echo "built $(realpath build/libtitok_hip_stamps64.so)"
