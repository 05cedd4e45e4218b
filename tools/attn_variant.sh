#!/bin/bash
# Diagnostic build: ttv_attn.hip with extra -D flags linked against the product objects -> csrc/build/libtitok_hip_<name>.so
#   tools/attn_variant.sh nolazy -DATTN_LAZY=0      then      TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_nolazy.so python tools/attn_bench.py
set -e
name=$1; shift
cd "$(dirname "$0")/../titok_video_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wall -Wno-unused-function"
hipcc $FLAGS "$@" -c ttv_attn.hip -o build/ttv_attn_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC build/ttv_elem.o build/ttv_gemm.o build/ttv_attn_$name.o build/ttv_attn_swp.o build/ttv_attn64.o build/ttv_mlp.o build/ttv_bwd.o build/ttv_train.o build/ttv_vq.o build/ttv_api.o -o build/libtitok_hip_$name.so
echo "built $(realpath build/libtitok_hip_$name.so)"
