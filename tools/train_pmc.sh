#!/bin/bash
# HBM traffic of the training step's kernels: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate, as the guide prescribes) over
# tools/bench_train.py (B = 32), summarised per kernel by tools/pmc_summarize.py -> gpurun_out/trainpmc/.  GPU box.
R=$(pwd); O=$R/gpurun_out/trainpmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STEPS=4 timeout -k 10 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/tools/bench_train.py > $O/f.txt 2> $O/f.log
echo "fetch pass: exit $?"
STEPS=4 timeout -k 10 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/tools/bench_train.py > $O/w.txt 2> $O/w.log
echo "write pass: exit $?"
cd $R
python3 tools/pmc_summarize.py --fetch $(find $O/f -name "*counter_collection.csv" | head -1) --write $(find $O/w -name "*counter_collection.csv" | head -1) --skip-first 8 --tag train --out $O | tail -2
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
