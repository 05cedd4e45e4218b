#!/bin/bash
# SQ counters of the inference forward (tiny, one chain)'s kernels (one pass: LDS bank conflicts, LDS instructions, MFMA busy cycles, wave cycles) over bench.py --config base5 --in-flight 1 --steps 3 --warmup 2 --no-cpu-baseline --no-fp32-leg.  GPU box.
R=$(pwd); O=$R/gpurun_out/base5sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/s -o s -- python3 $R/bench.py --config base5 --in-flight 1 --steps 3 --warmup 2 --no-cpu-baseline --no-fp32-leg > $O/s.txt 2> $O/s.log
echo "sq pass: exit $?"
python3 - $O/s <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    rows.append((len(next(iter(c.values()))) * m.get("SQ_BUSY_CYCLES", 0), k, len(next(iter(c.values()))), m))
rows.sort(reverse=True, key=lambda t: t[0])
print(f"{'kernel':70s} {'n':>4s} {'busy cyc':>12s} {'mfma busy':>12s} {'lds active':>12s} {'bank confl':>12s} confl/active")
for _, k, n, m in rows[:14]:
    a = m.get("SQ_LDS_IDX_ACTIVE", 0); b = m.get("SQ_LDS_BANK_CONFLICT", 0)
    print(f"{k[:70]:70s} {n:4d} {m.get('SQ_BUSY_CYCLES',0):12.0f} {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0):12.0f} {a:12.0f} {b:12.0f} {b/a if a else 0:6.3f}")
P
find $O -name "*kernel_trace.csv" -delete
