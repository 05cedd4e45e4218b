#!/usr/bin/env python3
"""Where a wave of k_attn_swp spends its cycles: per-segment s_memtime sums of the key loop plus prologue / epilogue (diagnostic build
tools/swp_stamps.sh, loaded through TTV_LIB_PATH).  Shares are meaningful, the run time of this build is not.

    bash tools/swp_stamps.sh && TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_swpstamps.so python3 tools/swp_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd import _lib  # noqa: E402
from titok_video_amd.plan import BatchPlan  # noqa: E402

DEV = torch.device("cuda:0")
lib = _lib.lib()
ST = _lib.stream_ptr(DEV)
B = int(os.environ.get("B", "32"))
plan = BatchPlan([(16, 128, 128)] * B, [128] * B, (4, 8, 8), DEV)
L = plan.total_rows
table = plan.attention_table(4, 2)
qkv = (torch.randn(L, 768, device=DEV) * (0.0 if os.environ.get("ZERO", "0") == "1" else 0.9)).to(torch.bfloat16)
qkv[:, :256] = (qkv[:, :256].float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
out = torch.empty(L, 256, dtype=torch.bfloat16, device=DEV)
n_samples = (table.shape[0] + 36) // 37
stamps = torch.zeros(n_samples * 4 * 8, dtype=torch.int64, device=DEV)
lib.ttv_debug_stamps(stamps.data_ptr())
NAMES = ["own DMA + fragment wait", "barrier", "X: S(t+1) | exp keys 32-63 | V reads | V DMA", "row sums (+ mask)", "Y: PV(t) | exp keys 0-31 | K reads | K DMA", "-"]
for _ in range(int(os.environ.get("REPS", "200"))):      # back to back: the clock the part settles at under this kernel
    _lib.check(lib.ttv_attention(qkv.data_ptr(), 768, out.data_ptr(), 256, plan.cu_dev.data_ptr(), table.data_ptr(), table.shape[0], 4, 2, 64, 1 | 4 | 8, 0, ST), "attn")
torch.cuda.synchronize()
s = stamps.view(n_samples, 4, 8).cpu().double()
seg = s[:, :, :6]
tiles = float(plan.total_rows // B // 64)
print(f"k_attn_swp, {B} x {plan.total_rows // B} rows, {table.shape[0]} entries: prologue {float(s[:, :, 6].mean()):.0f} cycles, loop {float(seg.sum(-1).mean()):.0f}, "
      f"epilogue {float(s[:, :, 7].mean()):.0f}  ({n_samples} sampled blocks x 4 waves)")
real = seg[:, :, 5].clone()
seg[:, :, 5] = 0
clk = seg.sum(-1) / real.clamp(min=1) * 0.1
print(f"  shader clock over the loop (s_memtime / s_memrealtime): {float(clk.mean()):.2f} GHz (min {float(clk.min()):.2f}, max {float(clk.max()):.2f})")
per_tile = seg / tiles
tot = per_tile.sum(-1)
for i, nm in enumerate(NAMES):
    v = per_tile[..., i]
    print(f"  {nm:38s} {float(v.mean()):8.0f} | {float(v.min()):8.0f} | {float(v.max()):8.0f}   {100 * float((v / tot).mean()):5.1f} %")
print(f"  {'total per tile and wave':38s} {float(tot.mean()):8.0f} | {float(tot.min()):8.0f} | {float(tot.max()):8.0f}   (a SIMD retires a unit every 1 / waves-per-SIMD of that)")
print(f"  first half of the grid {float(tot[: n_samples // 2].mean()):.0f}, second half {float(tot[n_samples // 2:].mean()):.0f} cycles per tile")
lib.ttv_debug_stamps(None)
