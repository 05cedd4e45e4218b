#!/bin/bash
# SQ counter passes over tools/qkv256_bench.py (both QKV kernels run in it) -> gpurun_out/<tag>/sq*.csv; summarise with tools/sq_summarize.py --match qkv / k_gemm_k256
tag=${1:-qkvsq}
O=gpurun_out/$tag; mkdir -p $O
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4; do
  case $i in 1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY";; 2) C="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA";; 3) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE";; 4) C="SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM";; esac
  echo "== pass $i: $C"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$O/sq$i -o s -- python3 $R/tools/qkv256_bench.py > /dev/null 2> $R/$O/sq$i.log || echo "pass $i failed"
done
cd $R
for k in k_qkv256 k_gemm_k256; do
  python3 tools/sq_summarize.py $(find $O -name "*counter_collection.csv") --match $k --skip-first 4 --out $O/sq_$k.csv
done
cat $O/sq_k_qkv256.csv $O/sq_k_gemm_k256.csv
