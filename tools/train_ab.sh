#!/bin/bash
# Round-5 training-step A/B on one box, alternating: the tree against (a) torch's clip + AdamW (TTV_HIP_ADAMW=0), (b) the erff() GEGLU backward
# (TTV_GEGLU_BWD_ERF=1), the one-row-per-wave RMSNorm, fp32 KEEL sums in the tape, (c) the round-4 weight-gradient kernel (csrc/build/libtitok_hip_r4bwd.so: also without (b); run with TTV_HIP_ADAMW=0 TTV_RMSNORM256=0 TTV_TAPE_Y_F32=1) -> gpurun_out/r05_train_ab.txt
O=gpurun_out/r05_train_ab.txt
echo "# tools/bench_train.py, 32 clips of 16x128x128, bf16, 20 steps each; alternating on one box" > $O
for r in 1 2 3; do
  echo "== tree" >> $O; STEPS=20 python tools/bench_train.py 2>/dev/null >> $O || exit 1
  echo "== tree, TTV_HIP_ADAMW=0 (torch clip_grad_norm_ + fused AdamW)" >> $O; STEPS=20 TTV_HIP_ADAMW=0 python tools/bench_train.py 2>/dev/null >> $O || exit 1
  echo "== tree, TTV_GEGLU_BWD_ERF=1" >> $O; STEPS=20 TTV_GEGLU_BWD_ERF=1 python tools/bench_train.py 2>/dev/null >> $O || exit 1
  echo "== tree, TTV_RMSNORM256=0 (one row per wave)" >> $O; STEPS=20 TTV_RMSNORM256=0 python tools/bench_train.py 2>/dev/null >> $O || exit 1
  echo "== tree, TTV_TAPE_Y_F32=1 (KEEL sums in fp32)" >> $O; STEPS=20 TTV_TAPE_Y_F32=1 python tools/bench_train.py 2>/dev/null >> $O || exit 1
  echo "== round-4 library (no loader waves, erff GEGLU backward), torch optimizer" >> $O; STEPS=20 TTV_HIP_ADAMW=0 TTV_RMSNORM256=0 TTV_TAPE_Y_F32=1 TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_r4bwd.so python tools/bench_train.py 2>/dev/null >> $O || exit 1
done
echo "== 5 clips (the reference's 6144-token budget): tree / round-4 library + torch optimizer" >> $O
for r in 1 2; do
  B=5 STEPS=30 python tools/bench_train.py 2>/dev/null >> $O || exit 1
  B=5 STEPS=30 TTV_HIP_ADAMW=0 TTV_RMSNORM256=0 TTV_TAPE_Y_F32=1 TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_r4bwd.so python tools/bench_train.py 2>/dev/null >> $O || exit 1
done
