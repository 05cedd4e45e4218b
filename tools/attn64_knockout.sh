#!/bin/bash
# Diagnostic builds of the 64-rows-per-wave attention kernel with single ingredients knocked out (garbage results, same launch):
# build/libtitok_hip_ko64_<name>.so for name in dma lds max exp all.  Run tools/attn_bench.py with TTV_LIB_PATH on each.
set -e
cd "$(dirname "$0")/../titok_video_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wall -Wno-unused-function"
for v in "dma:-DW64_KO_DMA=1" "lds:-DW64_KO_LDS=1" "max:-DW64_KO_MAX=1" "exp:-DW64_KO_EXP=1" "all:-DW64_KO_DMA=1 -DW64_KO_LDS=1 -DW64_KO_MAX=1 -DW64_KO_EXP=1"; do
  name=${v%%:*}; defs=${v#*:}
  bash build_attn64.sh build/ttv_attn64_ko.o $defs
  hipcc --offload-arch=gfx950 -shared -fPIC build/ttv_elem.o build/ttv_gemm.o build/ttv_attn.o build/ttv_attn64_ko.o build/ttv_mlp.o build/ttv_bwd.o build/ttv_train.o build/ttv_vq.o build/ttv_api.o -o build/libtitok_hip_ko64_$name.so
done
echo built
