#!/bin/bash
R=$(pwd); O=$R/gpurun_out/trainprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python3 $R/tools/bench_train.py > $O/bench.txt 2> $O/log.txt
python3 - $O/p <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ns", tot)
for r in rows[:45]:
    print(f'{r["Name"][:100]:100s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:8.1f} us {100*float(r["TotalDurationNs"])/tot:5.1f}%')
P
cat $O/bench.txt | tail -2
