#!/bin/bash
# Attention launch time against the number of table entries (full items only): how the time grows with the blocks per CU.   GPU box.
for b in 3 7 14 21 28 32 42 64; do echo "== B=$b (items $((b*36)))"; TTV_ATTN_SPLIT=0 TTV_ATTN_PERS=${PERS:-0} B=$b FP32=0 timeout -k 10 120 python3 tools/attn_bench.py 1.5 2>&1 | grep "gate+qscaled  "; done
