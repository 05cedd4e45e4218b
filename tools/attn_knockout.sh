#!/bin/bash
# Diagnostic builds of the pipelined attention kernel with one ingredient knocked out each (results are garbage, the launch time
# tells what the ingredient costs): build/libtitok_hip_ko_{dma,barrier,valu,lds,all}.so;  run with
#   TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_ko_dma.so FP32=0 python tools/attn_bench.py 1.5
set -e
cd "$(dirname "$0")/../titok_video_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wall -Wno-unused-function"
for v in dma barrier valu lds all; do
  case $v in
    dma) D="-DPP_KO_DMA=1";; barrier) D="-DPP_KO_BARRIER=1";; valu) D="-DPP_KO_VALU=1";; lds) D="-DPP_KO_LDS=1";; all) D="-DPP_KO_DMA=1 -DPP_KO_BARRIER=1 -DPP_KO_VALU=1 -DPP_KO_LDS=1";;
  esac
  hipcc $FLAGS $D -c ttv_attn.hip -o build/ttv_attn_ko_$v.o
  hipcc --offload-arch=gfx950 -shared -fPIC build/ttv_elem.o build/ttv_gemm.o build/ttv_attn_ko_$v.o build/ttv_attn64.o build/ttv_mlp.o build/ttv_bwd.o build/ttv_train.o build/ttv_vq.o build/ttv_api.o -o build/libtitok_hip_ko_$v.so
done
echo "built knock-out libraries"
