#!/usr/bin/env python3
"""Per-segment s_memtime sums of the to_qkv kernel (k_qkv256, -DQKV_STAMPS build: tools/qkv256_stamps.sh) at the benchmark shape.
Every 16th block reports, per wave, the cycles its items spent in: token-tile change | first-panel wait | DMA issue | 32 MFMAs +
riding epilogue | stores | counted wait | barrier.  Printed: mean per ITEM over the reporting waves, and the spread."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd import _lib  # noqa: E402
from titok_video_amd.plan import BatchPlan  # noqa: E402

DEV = torch.device("cuda:0")
lib = _lib.lib()
S = _lib.stream_ptr(DEV)
B = int(os.environ.get("B", "32"))
plan = BatchPlan([(16, 128, 128)] * B, [128] * B, (4, 8, 8), DEV)
L, d, g = plan.total_rows, 256, 128
bf = torch.bfloat16
x = torch.randn(L, d, device=DEV).to(bf)
w = (torch.randn(2 * d + 2 * g, d, device=DEV) * d ** -0.5).to(bf)
qkv = torch.empty(L, 2 * d + 2 * g, dtype=bf, device=DEV)


def call():
    _lib.check(lib.ttv_linear_qkv_rope(x.data_ptr(), d, w.data_ptr(), d, qkv.data_ptr(), 2 * d + 2 * g, L, d, g, plan.rope_cs.data_ptr(), 0, S), "qkv")


for _ in range(20):
    call()
torch.cuda.synchronize()
st = torch.zeros(64 * 4 * 8, dtype=torch.int64, device=DEV)
lib.ttv_debug_stamps(st.data_ptr())
call()
torch.cuda.synchronize()
lib.ttv_debug_stamps(None)
v = st.cpu().view(-1, 8)
v = v[v[:, 7] > 0].double()
names = ["tile change", "first wait", "DMA issue", "MFMA+epilogue", "stores", "counted wait", "barrier"]
items = v[:, 7]
print(f"{len(v)} waves reporting, items per wave {items.min():.0f}..{items.max():.0f}")
tot = v[:, :7].sum(1)
print(f"cycles per wave inside the item loop: mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}")
for i, n in enumerate(names):
    per = v[:, i] / items
    print(f"  {n:14s} per item: mean {per.mean():8.1f}  min {per.min():8.1f}  max {per.max():8.1f}   ({100 * v[:, i].sum() / tot.sum():5.1f} %)")
