#!/usr/bin/env python3
"""Shader clock held under the weight-stationary to_qkv kernel (-DQKV_STAMPS builds of tools/qkv256_knockout.sh): per reporting wave the
s_memtime cycles and the 100 MHz s_memrealtime ticks of (kernel entry -> weights landed) and of the unit loop.  `debug` 1 = no stores."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd import _lib  # noqa: E402
from titok_video_amd.plan import BatchPlan  # noqa: E402

DEV = torch.device("cuda:0")
lib = _lib.lib()
S = _lib.stream_ptr(DEV)
plan = BatchPlan([(16, 128, 128)] * 32, [128] * 32, (4, 8, 8), DEV)
L, d, g = plan.total_rows, 256, 128
bf = torch.bfloat16
x = torch.randn(L, d, device=DEV).to(bf)
w = (torch.randn(2 * d + 2 * g, d, device=DEV) * d ** -0.5).to(bf)
qkv = torch.empty(L, 2 * d + 2 * g, dtype=bf, device=DEV)


def call():
    _lib.check(lib.ttv_linear_qkv_rope(x.data_ptr(), d, w.data_ptr(), d, qkv.data_ptr(), 2 * d + 2 * g, L, d, g, plan.rope_cs.data_ptr(), 0, S), "qkv")


for dbg in (0, 1):
    lib.ttv_debug_set(dbg)
    for _ in range(200):       # the clock settles over many back-to-back launches
        call()
    torch.cuda.synchronize()
    st = torch.zeros(8 * 8 * 8 * 8, dtype=torch.int64, device=DEV)
    lib.ttv_debug_stamps(st.data_ptr())
    call()
    torch.cuda.synchronize()
    lib.ttv_debug_stamps(None)
    v = st.cpu().view(-1, 8).double()
    v = v[v[:, 7] > 0]
    for name, c, r in (("entry -> weights landed", 0, 1), ("unit loop", 2, 3)):
        cyc, us = v[:, c], v[:, r] / 100.0
        print(f"debug {dbg} {name:24s}: {len(v)} waves, {cyc.mean():9.0f} cycles (max {cyc.max():.0f}), {us.mean():6.2f} us (max {us.max():.2f}) -> {cyc.sum() / us.sum() / 1e3:.2f} GHz; "
              f"items per wave {v[:, 7].mean():.2f}, cycles per item {(v[:, 2] / v[:, 7]).mean():.0f}" if c == 2 else
              f"debug {dbg} {name:24s}: {len(v)} waves, {cyc.mean():9.0f} cycles (max {cyc.max():.0f}), {us.mean():6.2f} us (max {us.max():.2f}) -> {cyc.sum() / us.sum() / 1e3:.2f} GHz", flush=True)
lib.ttv_debug_set(0)
