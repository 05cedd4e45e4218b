#!/bin/bash
# The split plan of the weight-gradient GEMMs (blocks aimed at, fewest 64-token steps per block) against the training step, B = 5 and 32.  GPU box.
for B in 5 32; do
  for cfg in "384 1" "384 8" "384 16" "256 8" "256 16" "192 12" "128 16" "512 8"; do
    set -- $cfg
    echo -n "B=$B blocks=$1 min_steps=$2: "
    B=$B TTV_WGRAD_BLOCKS=$1 TTV_WGRAD_MIN_STEPS=$2 timeout -k 5 60 python tools/bench_train.py 2>/dev/null | tail -1 | cut -c1-60
  done
done
