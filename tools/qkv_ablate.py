#!/usr/bin/env python3
"""Knock-outs of the pre-norm + to_qkv + rotary kernel (k_gemm_k256<EPI_QKV_ROPE, true>) at the benchmark shape (36 864 tokens,
K = 256, N = 768): ttv_debug_set bits 1 = no stores (and no rotary), 2 = no panel DMA after the first, 4 = no token-tile reload,
8 = no epilogue arithmetic (rstd scale, rotary).  Garbage results under any bit; timing only.  GPU box."""
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


def t(fn, it=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


def call():
    _lib.check(lib.ttv_linear_qkv_rope(x.data_ptr(), d, w.data_ptr(), d, qkv.data_ptr(), 2 * d + 2 * g, L, d, g, plan.rope_cs.data_ptr(), 0, S), "qkv")


for dbg in (0, 1, 2, 4, 8, 1 | 8, 2 | 4, 1 | 2 | 4, 1 | 2 | 4 | 8, 0):
    lib.ttv_debug_set(dbg)
    print(f"debug {dbg:3d}: {t(call):7.1f} us", flush=True)
lib.ttv_debug_set(0)
