#!/bin/bash
# The three to_qkv kernels inside the benchmark forward (folded pre-norm, rotary factors by position id): per-kernel averages from
# rocprofv3 --kernel-trace --stats of `bench.py --in-flight 1`, once per TTV_QKV256 = 0 (k_gemm_k256) / 1 (k_qkv256) / 2 (k_qkv256ws).
tag=${1:-qkvpipe}
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in ${MODES:-0 1 2}; do
  TTV_QKV256=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m$m -o p -- python3 $R/bench.py --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-side-legs > $O/bench_m$m.json 2> $O/m$m.log
  echo "== TTV_QKV256=$m"
  python3 - $O/m$m <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:7]:
    print(f'  {r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  {100*float(r["TotalDurationNs"])/tot:5.1f} %')
P
  python3 -c "import json,sys; d=json.loads([l for l in open('$O/bench_m$m.json') if l.startswith('{')][-1]); print('  one chain under rocprof: %.1f clips/s, %.4f ms/step' % (d['value'], d['ms_per_step']))"
done
