#!/bin/bash
# Diagnostic build: ttv_mlp.hip with extra -D flags linked against the product objects -> csrc/build/libtitok_hip_<name>.so
#   tools/mlp_variant.sh depth2 -DMLP_P1_DEPTH=2     then     TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_depth2.so python tools/mlp_ablate.py
set -e
name=$1; shift
cd "$(dirname "$0")/../titok_video_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wall -Wno-unused-function"
hipcc $FLAGS "$@" -c ttv_mlp.hip -o build/ttv_mlp_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC build/ttv_elem.o build/ttv_gemm.o build/ttv_attn.o build/ttv_attn_swp.o build/ttv_attn64.o build/ttv_mlp_$name.o build/ttv_bwd.o build/ttv_train.o build/ttv_vq.o build/ttv_api.o -o build/libtitok_hip_$name.so
echo "built $(realpath build/libtitok_hip_$name.so)"
