#!/bin/bash
# A/B library: the tree's objects with ttv_bwd.hip as it was before round 5's weight-gradient / GEGLU-backward work (git a48f7fb~1)
#   -> csrc/build/libtitok_hip_r4bwd.so   (used by tools/train_ab.sh, tools/wgrad_ab.sh through TTV_LIB_PATH)
set -e
cd "$(dirname "$0")/../titok_video_amd/csrc"
git show a48f7fb~1:titok_video_amd/csrc/ttv_bwd.hip > build/ttv_bwd_r4.hip
# the one interface that changed since: ttvk_rmsnorm_bwd_chain takes the dtype of y (the old source reads fp32: run with TTV_TAPE_Y_F32=1)
python3 - <<'P'
p = "build/ttv_bwd_r4.hip"
s = open(p).read()
old = """int ttvk_rmsnorm_bwd_chain(const void* x, int ldx, const void* dy, int lddy, const float* gain1, float* dgain1, float* dx, int lddx,
                           const float* y, int ldy, const float* gain2,"""
new = """int ttvk_rmsnorm_bwd_chain(const void* x, int ldx, const void* dy, int lddy, const float* gain1, float* dgain1, float* dx, int lddx,
                           const void* y_, int y_dt, int ldy, const float* gain2,"""
assert old in s
s = s.replace(old, new, 1)
i = s.index(new)
j = s.index("{", s.index("hipStream_t s)", i)) + 1
s = s[:j] + "\n  const float* y = (const float*)y_;\n  TTV_CHECK_ARG(!y || y_dt == TTV_F32, \"round-4 backward: fp32 KEEL sums only (TTV_TAPE_Y_F32=1)\");" + s[j:]
open(p, "w").write(s)
P
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wall -Wno-unused-function -I. -I../../include -c build/ttv_bwd_r4.hip -o build/ttv_bwd_r4.o
hipcc --offload-arch=gfx950 -shared -fPIC build/ttv_elem.o build/ttv_gemm.o build/ttv_attn.o build/ttv_attn_swp.o build/ttv_attn64.o build/ttv_mlp.o build/ttv_bwd_r4.o build/ttv_train.o build/ttv_vq.o build/ttv_api.o -o build/libtitok_hip_r4bwd.so
echo "built $(realpath build/libtitok_hip_r4bwd.so)"
