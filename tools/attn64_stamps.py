#!/usr/bin/env python3
"""Where a wave of the 64-rows-per-wave attention kernel (ttv_attention64) spends its cycles: per-segment s_memtime sums of the key
loop, prologue and epilogue (diagnostic build tools/attn64_stamps.sh, loaded through TTV_LIB_PATH).  Shares are meaningful, the run
time of this build is not (the stamps fence the scheduler and cost ~10 %).

    bash tools/attn64_stamps.sh && TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_stamps64.so python3 tools/attn64_stamps.py"""
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
CLIP = tuple(int(v) for v in os.environ.get("CLIP", "16,128,128").split(","))
KTOK = int(os.environ.get("K", "128"))
HQ, HKV = int(os.environ.get("HQ", "4")), int(os.environ.get("HKV", "2"))
plan = BatchPlan([CLIP] * B, [KTOK] * B, (4, 8, 8), DEV)
L = plan.total_rows
S = L // B
dm, g = HQ * 64, HKV * 64
ld = 2 * dm + 2 * g
table = plan.attention_table64(HQ, HKV)
qkv = (torch.randn(L, ld, device=DEV) * 0.9).to(torch.bfloat16)
qkv[:, :dm] = (qkv[:, :dm].float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
out = torch.empty(L, dm, dtype=torch.bfloat16, device=DEV)
n_samples = (table.shape[0] + 36) // 37
stamps = torch.zeros(n_samples * 4 * 8, dtype=torch.int64, device=DEV)
lib.ttv_debug_stamps(stamps.data_ptr())
NAMES = ["DMA wait (vmcnt) + barrier", "Ya x2: 4 PV_B | rowmax A (+1 DMA)", "Yb x2: 4 S_B | exp A + V reads", "Xa x2: 4 PV_A | rowmax B + K reads (+1 DMA)",
         "Xb x2: 4 S_A | exp B", "-"]
for _ in range(3):
    _lib.check(lib.ttv_attention64(qkv.data_ptr(), ld, out.data_ptr(), dm, plan.cu_dev.data_ptr(), table.data_ptr(), table.shape[0], HQ, HKV, 64, 1 | 4, 0, ST),
               "attn64")
torch.cuda.synchronize()
s = stamps.view(n_samples, 4, 8).cpu().double()
nkt = (S + 63) // 64
seg = s[:, :, :6] / nkt
tot = seg.sum(-1)
print(f"{B} x {S} rows, {HQ}/{HKV} heads: {table.shape[0]} workgroups, {nkt} key tiles; {n_samples} sampled workgroups x 4 waves")
print(f"prologue {float(s[:, :, 6].mean()):.0f} cycles, loop {float(s[:, :, :6].sum(-1).mean()):.0f}, epilogue {float(s[:, :, 7].mean()):.0f}")
print("cycles per key tile and wave (64 query rows: 32 MFMAs = 1024 matrix-pipe cycles): mean | min | max, share")
for i, nm in enumerate(NAMES):
    v = seg[..., i]
    print(f"  {nm:44s} {float(v.mean()):8.0f} | {float(v.min()):8.0f} | {float(v.max()):8.0f}   {100 * float((v / tot).mean()):5.1f} %")
print(f"  {'total per tile':44s} {float(tot.mean()):8.0f} | {float(tot.min()):8.0f} | {float(tot.max()):8.0f}")
lib.ttv_debug_stamps(None)
