#!/bin/bash
# Knock-out builds of the weight-stationary to_qkv kernel (ttv_qkv256ws.inc: -DQW_KO_LDS / _EPI / _MFMA / _X) -> titok_video_amd/csrc/variants/*.so
#   tools/qkv256_knockout.sh build   (here: hipcc cross-compiles; the .so files travel to the GPU box)
#   tools/qkv256_knockout.sh run     (GPU box: tools/qkv256_bench.py under every variant)
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
V=$R/titok_video_amd/csrc/variants
NAMES=${NAMES:-"base lds epi mfma x lds_epi lds_epi_x rows"}
flags() { case $1 in base) echo "";; lds) echo "-DQW_KO_LDS=1";; epi) echo "-DQW_KO_EPI=1";; mfma) echo "-DQW_KO_MFMA=1";; x) echo "-DQW_KO_X=1";; lds_epi) echo "-DQW_KO_LDS=1 -DQW_KO_EPI=1";; lds_epi_x) echo "-DQW_KO_LDS=1 -DQW_KO_EPI=1 -DQW_KO_X=1";; rows) echo "-DQW_KO_ROWS=1";; st_base) echo "-DQKV_STAMPS";; st_mfmaonly) echo "-DQKV_STAMPS -DQW_KO_LDS=1 -DQW_KO_EPI=1 -DQW_KO_X=1";; esac; }
if [ "$1" = build ]; then
  mkdir -p $V; cd $R/titok_video_amd/csrc
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wno-unused-function -Wno-unused-variable"
  for n in $NAMES; do
    ( hipcc $FLAGS $(flags $n) -c ttv_gemm.hip -o $V/gemm_$n.o 2> $V/gemm_$n.log &&
      hipcc --offload-arch=gfx950 -shared -fPIC build/ttv_elem.o $V/gemm_$n.o build/ttv_attn.o build/ttv_attn_swp.o build/ttv_attn64.o build/ttv_mlp.o build/ttv_bwd.o build/ttv_train.o build/ttv_vq.o build/ttv_api.o -o $V/libtitok_hip_qw_$n.so && rm $V/gemm_$n.o ) &
    [ $(jobs -r | wc -l) -ge 4 ] && wait -n
  done
  wait; ls -la $V/*.so
else
  for n in $NAMES; do
    echo "== variant $n"
    case $n in st_*) TTV_LIB_PATH=$V/libtitok_hip_qw_$n.so python3 $R/tools/qkv256ws_clock.py 2>&1 | grep -v amdgpu.ids;;
      *) TTV_LIB_PATH=$V/libtitok_hip_qw_$n.so ONLY_WS=1 python3 $R/tools/qkv256_bench.py 2>&1 | grep -v amdgpu.ids | grep "k_qkv256ws" | tail -4;; esac
  done
fi
