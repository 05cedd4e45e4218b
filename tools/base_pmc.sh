#!/bin/bash
# HBM traffic of the base workload's kernels (BASELINE config #4): FETCH_SIZE / WRITE_SIZE passes over bench.py --config base, one chain.  GPU box.
R=$(pwd); O=$R/gpurun_out/basepmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/bench.py --config base --in-flight 1 --steps 3 --warmup 2 --no-cpu-baseline --no-fp32-leg > $O/f.txt 2> $O/f.log
echo "fetch pass: exit $?"
timeout -k 10 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/bench.py --config base --in-flight 1 --steps 3 --warmup 2 --no-cpu-baseline --no-fp32-leg > $O/w.txt 2> $O/w.log
echo "write pass: exit $?"
cd $R
python3 tools/pmc_summarize.py --fetch $(find $O/f -name "*counter_collection.csv" | head -1) --write $(find $O/w -name "*counter_collection.csv" | head -1) --skip-first 4 --tag base --out $O | tail -1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
