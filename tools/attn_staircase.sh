for b in 16 21 22 32 42 43 48 64; do echo "== B=$b (items $((b*36)))"; B=$b FP32=0 timeout -k 10 120 python3 tools/attn_bench.py 1.5 2>&1 | grep "gate+qscaled  \|gate+qscaled w64"; done
