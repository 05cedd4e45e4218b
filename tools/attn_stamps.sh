#!/bin/bash
# Diagnostic build: the attention and layer-tail sources with -DATTN_STAMPS / -DMLP_STAMPS linked against the product objects -> build/libtitok_hip_stamps.so
set -e
cd "$(dirname "$0")/../titok_video_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wall -Wno-unused-function"
hipcc $FLAGS -DATTN_STAMPS -c ttv_attn.hip -o build/ttv_attn_stamps.o
hipcc $FLAGS -DMLP_STAMPS -c ttv_mlp.hip -o build/ttv_mlp_stamps.o
hipcc --offload-arch=gfx950 -shared -fPIC build/ttv_elem.o build/ttv_gemm.o build/ttv_attn_stamps.o build/ttv_attn_swp.o build/ttv_attn64.o build/ttv_mlp_stamps.o build/ttv_bwd.o build/ttv_train.o build/ttv_vq.o build/ttv_api.o -o build/libtitok_hip_stamps.so
echo "built $(realpath build/libtitok_hip_stamps.so)"
