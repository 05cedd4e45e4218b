#!/bin/bash
# SQ counter passes over tools/wgrad_bench.py -> gpurun_out/<tag>/sq*.csv, summarised per kernel class by tools/sq_summarize.py
#   tools/wgrad_counters.sh wgsq        (TTV_LIB_PATH picks the library as everywhere)
tag=${1:-wgsq}
O=gpurun_out/$tag; mkdir -p $O
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4; do
  case $i in 1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY";; 2) C="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA";; 3) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE";; 4) C="SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM";; esac
  echo "== pass $i: $C"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$O/sq$i -o s -- python3 $R/tools/wgrad_bench.py > /dev/null 2> $R/$O/sq$i.log || echo "pass $i failed"
done
cd $R
python3 tools/sq_summarize.py $(find $O -name "*counter_collection.csv") --match k_wgrad128 --skip-first 4 --out $O/sq_k_wgrad128.csv
cat $O/sq_k_wgrad128.csv
