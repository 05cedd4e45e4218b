#!/bin/bash
# Diagnostic build: ttv_bwd.hip with -DWG_STAMPS (extra -D flags pass through) linked against the product objects -> build/libtitok_hip_wgstamps.so
set -e
cd "$(dirname "$0")/../titok_video_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wall -Wno-unused-function"
hipcc $FLAGS -DWG_STAMPS "$@" -c ttv_bwd.hip -o build/ttv_bwd_wgstamps.o
hipcc --offload-arch=gfx950 -shared -fPIC build/ttv_elem.o build/ttv_gemm.o build/ttv_attn.o build/ttv_attn_swp.o build/ttv_attn64.o build/ttv_mlp.o build/ttv_bwd_wgstamps.o build/ttv_train.o build/ttv_vq.o build/ttv_api.o -o build/libtitok_hip_wgstamps.so
echo "built $(realpath build/libtitok_hip_wgstamps.so)"
