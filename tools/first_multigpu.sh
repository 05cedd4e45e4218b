#!/bin/bash
# ONE command for the first lease of a box with more than one GPU.  Nothing here has ever run on more than one
# GPU: RCCL (torch.distributed backend "nccl"), the record_stream choreography of dp.GradReducer under a real process group and
# bench.py --gpus N > 1 are covered by gloo tests on the CPU and by a one-GPU `simulate` mode only.  Every step is a FRESH process tree
# (bench.py / train_dp.py fan out their own ranks before touching the GPU; nothing re-execs a process that has initialised HIP), runs
# under its own timeout, and writes into profiles/.  A step that times out stops the script (no further GPU step on a box that may be
# unhealthy); an ordinary failure is recorded and the next step still runs.
#     bash tools/first_multigpu.sh [tag]        (from the repository root; needs >= 2 visible GPUs)
set -u
TAG=${1:-r04}
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
N=$(python3 -c 'import torch; print(torch.cuda.device_count())')     # device_count() does not initialise the GPU
if [ "$N" -lt 2 ]; then echo "first_multigpu.sh: $N GPU(s) visible - nothing to do (this script is for the first multi-GPU box)"; exit 2; fi
mkdir -p profiles
LOG=profiles/${TAG}_first_multigpu.log
: > "$LOG"
step() {            # step <name> <seconds> <output file | -> <command...>
  local name=$1 secs=$2 out=$3; shift 3
  echo "== $name: $*" | tee -a "$LOG"
  if [ "$out" = "-" ]; then timeout -k 10 "$secs" "$@" >> "$LOG" 2>&1; else timeout -k 10 "$secs" "$@" > "$out" 2>> "$LOG"; fi
  local rc=$?
  echo "   -> exit $rc" | tee -a "$LOG"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name timed out / was killed: stopping" | tee -a "$LOG"; exit $rc; fi
}
# 1. the RCCL test once (2 ranks): gradient all-reduce against the single-process reference
step rccl_test 600 - python3 -m pytest tests/test_hip_dp_train.py -q -m gpu -k rccl -s
# 2. the headline bench, weak scaling: bench.py starts its own ranks
for g in 2 4 8; do
  [ "$g" -le "$N" ] || continue
  step bench_gpus$g 900 profiles/${TAG}_bench_gpus$g.json python3 bench.py --gpus $g --no-cpu-baseline --no-fp32-leg
done
# 3. BASELINE config #3: DP training on synthetic shards (overlapped gradient all-reduce, codebook histogram all-reduce)
G=$N; [ "$G" -gt 8 ] && G=8
step train_dp_gpus$G 900 profiles/${TAG}_train_dp_gpus$G.json python3 tools/train_dp.py --gpus $G --steps 100 --warmup 10
step train_dp_gpus${G}_no_overlap 900 profiles/${TAG}_train_dp_gpus${G}_no_overlap.json python3 tools/train_dp.py --gpus $G --steps 100 --warmup 10 --no-overlap
echo "first_multigpu.sh: done; results in profiles/${TAG}_*gpus*.json, log $LOG"
