#!/bin/bash
# A/B library: the tree's objects with ttv_bwd.hip as it was before round 5's weight-gradient / GEGLU-backward work (git a48f7fb~1)
#   -> csrc/build/libtitok_hip_r4bwd.so   (used by tools/train_ab.sh, tools/wgrad_ab.sh through TTV_LIB_PATH)
set -e
cd "$(dirname "$0")/../titok_video_amd/csrc"
git show a48f7fb~1:titok_video_amd/csrc/ttv_bwd.hip > build/ttv_bwd_r4.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wall -Wno-unused-function -I. -I../../include -c build/ttv_bwd_r4.hip -o build/ttv_bwd_r4.o
hipcc --offload-arch=gfx950 -shared -fPIC build/ttv_elem.o build/ttv_gemm.o build/ttv_attn.o build/ttv_attn_swp.o build/ttv_attn64.o build/ttv_mlp.o build/ttv_bwd_r4.o build/ttv_train.o build/ttv_vq.o build/ttv_api.o -o build/libtitok_hip_r4bwd.so
echo "built $(realpath build/libtitok_hip_r4bwd.so)"
