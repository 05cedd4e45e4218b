#!/bin/bash
# Diagnostic build of the to_qkv kernel with -DQKV_STAMPS (ttv_qkv256.inc) -> build/libtitok_hip_qkvstamps.so, then tools/qkv256_stamps.py.
# The object files do not travel to the GPU box (.gpurunignore): the product objects are rebuilt there first.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R/titok_video_amd/csrc"
[ -f build/ttv_api.o ] || bash build.sh > /dev/null 2>&1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wno-unused-function -Wno-unused-variable"
hipcc $FLAGS -DQKV_STAMPS $QKV_EXTRA -c ttv_gemm.hip -o build/ttv_gemm_qkvstamps.o
hipcc --offload-arch=gfx950 -shared -fPIC build/ttv_elem.o build/ttv_gemm_qkvstamps.o build/ttv_attn.o build/ttv_attn_swp.o build/ttv_attn64.o build/ttv_mlp.o build/ttv_bwd.o build/ttv_train.o build/ttv_vq.o build/ttv_api.o -o build/libtitok_hip_qkvstamps.so
cd "$R"
TTV_LIB_PATH=$R/titok_video_amd/csrc/build/libtitok_hip_qkvstamps.so python3 tools/qkv256_stamps.py
