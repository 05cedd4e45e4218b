#!/bin/bash
# rocprofv3 kernel stats of tools/bench_gan_train.py (B= batch, default 5): per-step totals by kernel -> stdout
R=$(pwd); O=$R/gpurun_out/ganprof_${B:-5}; mkdir -p $O
export B=${B:-5}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python3 $R/tools/bench_gan_train.py > $O/bench.txt 2> $O/log.txt
python3 - $O/p <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 13
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot/steps/1e6:.3f} ms, launches per step {sum(int(r['Calls']) for r in rows)/steps:.0f}")
for r in rows[:40]:
    print(f'{r["Name"][:90]:90s} {int(r["Calls"])/steps:6.1f}/step {float(r["AverageNs"])/1e3:8.1f} us {float(r["TotalDurationNs"])/steps/1e3:8.1f} us/step {100*float(r["TotalDurationNs"])/tot:5.1f}%')
P
tail -1 $O/bench.txt
