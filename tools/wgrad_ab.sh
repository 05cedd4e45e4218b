O=gpurun_out/r05_wgrad_loader_waves.txt
echo "# weight-gradient GEMM with loader waves (tree) against the round-4 kernel (csrc/build/libtitok_hip_r4bwd.so from tools/r4bwd_lib.sh; run as TTV_TAPE_Y_F32=1 bash tools/wgrad_ab.sh - the old source reads fp32 KEEL sums), same box, alternating" > $O
for v in tree prev; do
  echo "== tools/wgrad_bench.py, $v" >> $O
  if [ $v = prev ]; then export TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_r4bwd.so; else unset TTV_LIB_PATH; fi
  python tools/wgrad_bench.py 2>/dev/null | grep wgrad >> $O || exit 1
done
for r in 1 2 3; do for v in tree prev; do
  if [ $v = prev ]; then export TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_r4bwd.so; else unset TTV_LIB_PATH; fi
  for sd in 1 0; do echo "== tools/bench_train.py (32 clips, 20 steps), $v, TTV_WGRAD_SIDE=$sd" >> $O; STEPS=20 TTV_WGRAD_SIDE=$sd python tools/bench_train.py 2>/dev/null >> $O || exit 1; done
done; done
unset TTV_LIB_PATH
echo "== B=5 (the reference's token budget), tree / prev" >> $O
B=5 STEPS=20 python tools/bench_train.py 2>/dev/null >> $O && B=5 STEPS=20 TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_r4bwd.so python tools/bench_train.py 2>/dev/null >> $O
echo "== stamps (tools/wgrad_stamps.sh build: every stamp drains the wave's LDS queue, ~50 cycles each, three per stage)" >> $O
if [ -f titok_video_amd/csrc/build/libtitok_hip_wgstamps.so ]; then TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_wgstamps.so python3 tools/wgrad_stamps.py 2>/dev/null >> $O; else echo "(tools/wgrad_stamps.sh not built)" >> $O; fi
