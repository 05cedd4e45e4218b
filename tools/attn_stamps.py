#!/usr/bin/env python3
"""Where a wave of the bf16 attention kernel spends its cycles: per-segment s_memtime sums of the key loop (diagnostic build
tools/attn_stamps.sh, loaded through TTV_LIB_PATH).  Shares are meaningful, the run time of this build is not.

    bash tools/attn_stamps.sh && TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_stamps.so python3 tools/attn_stamps.py"""
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
qkv = (torch.randn(L, 768, device=DEV) * 0.9).to(torch.bfloat16)
out = torch.empty(L, 256, dtype=torch.bfloat16, device=DEV)
n_samples = (table.shape[0] + 36) // 37
stamps = torch.zeros(n_samples * 4 * 8, dtype=torch.int64, device=DEV)
lib.ttv_debug_stamps(stamps.data_ptr())
NAMES = {"plain": ["own DMA wait", "barrier", "DMA issue+K reads+S mfma issue", "S done+max+exchange", "exp2+pack", "V reads+PV issue"],
         "pipe": ["step A: PV | softmax 0-15", "step A: S | softmax 16-31 + check", "own DMA wait", "barrier", "DMA issue", "step B (both blocks)"]}
for flags, name in ((1, "gate"), (1 | 4, "gate+qscaled"), (1 | 4 | 8 | 16, "gate+qscaled pipe")):
    x = qkv.clone()
    if flags & 4:      # pre-scaled q: the factor head_dim^-0.5 * log2(e) is in the data
        x[:, :256] = (x[:, :256].float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
    for _ in range(3):
        _lib.check(lib.ttv_attention(x.data_ptr(), 768, out.data_ptr(), 256, plan.cu_dev.data_ptr(), table.data_ptr(), table.shape[0], 4, 2, 64, flags, 0, ST), "attn")
    torch.cuda.synchronize()
    s = stamps.view(n_samples, 4, 8).cpu().double()
    seg = s[:, :, :6]
    tiles = s[:, :, 6].clamp(min=1)
    if flags & 8:      # the pipelined kernel reports entry -> loop and loop -> end instead of the tile count
        tiles = torch.full_like(tiles, float(plan.total_rows // B // 64))
        print(f"---- {name}: prologue {float(s[:, :, 6].mean()):.0f} cycles, loop {float(seg.sum(-1).mean()):.0f}, epilogue {float(s[:, :, 7].mean()):.0f} "
              f"(first half of the grid: {float(s[:n_samples // 2, :, 6].mean()):.0f} / {float(seg[:n_samples // 2].sum(-1).mean()):.0f} / {float(s[:n_samples // 2, :, 7].mean()):.0f})")
    per_tile = (seg / tiles[..., None])
    tot = per_tile.sum(-1)
    names = NAMES["pipe" if flags & 8 else "plain"]
    print(f"---- {name}: {n_samples} sampled blocks x 4 waves; cycles per key tile (mean | min | max over sampled waves)")
    for i, nm in enumerate(names):
        v = per_tile[..., i]
        print(f"  {nm:34s} {float(v.mean()):8.0f} | {float(v.min()):8.0f} | {float(v.max()):8.0f}   {100 * float((v / tot).mean()):5.1f} %")
    print(f"  {'total per tile':34s} {float(tot.mean()):8.0f} | {float(tot.min()):8.0f} | {float(tot.max()):8.0f}")
    first = tot[: n_samples // 2].mean()
    last = tot[n_samples // 2:].mean()
    print(f"  first half of the grid {float(first):.0f}, second half {float(last):.0f} cycles per tile")
lib.ttv_debug_stamps(None)
