#!/usr/bin/env python3
"""Attention backward alone at the training-step shape (B sequences x 1152 rows, 4 q-heads / 2 kv-heads): k_attn_bwd (dK/dV and dQ blocks in
one grid) back to back, by torch events.  GPU box only.    B=32 python tools/attn_bwd_bench.py"""
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
hq, hkv, d, gq = 4, 2, 256, 128
ld = 2 * d + 2 * gq
L = plan.total_rows
g = torch.Generator().manual_seed(7)
qkvg = (torch.randn(L, ld, generator=g) * 0.7).to(DEV, torch.bfloat16)
dout = torch.randn(L, d, generator=g).to(DEV, torch.bfloat16)
o = torch.empty(L, d, dtype=torch.bfloat16, device=DEV)
lse = torch.empty(L, hq, device=DEV)
tab = plan.attention_table(hq, hkv)
code = _lib.TTV_BF16
_lib.check(lib.ttv_attention_lse(qkvg.data_ptr(), ld, o.data_ptr(), d, plan.cu_dev.data_ptr(), tab.data_ptr(), tab.shape[0], hq, hkv, 64, 0, code,
                                 lse.data_ptr(), ST), "attention_lse")
dq = torch.zeros(L, ld, dtype=torch.bfloat16, device=DEV)
delta = torch.empty(L, hq, device=DEV)
scratch = torch.empty(L, 2 * gq, device=DEV)
bt = plan.table(4, 2 * plan.n_blocks64)
rs = plan.table(5, L)


def call():
    _lib.check(lib.ttv_attention_backward(qkvg.data_ptr(), ld, o.data_ptr(), d, dout.data_ptr(), d, lse.data_ptr(), delta.data_ptr(), plan.cu_dev.data_ptr(),
                                          bt.data_ptr(), plan.n_blocks64, rs.data_ptr(), dq.data_ptr(), ld, scratch.data_ptr(), L, hq, hkv, code,
                                          plan.rope_cs.data_ptr(), ST), "attention_backward")


for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    call()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
S = L // B
flops5 = B * hq * 5 * 2.0 * S * S * 64           # the five products of the textbook backward
print(f"attention backward {B} x {S} rows (delta + k_attn_bwd): {us:7.1f} us per call, {flops5 / us / 1e6:6.1f} TFLOP/s on the 5-product count "
      f"({flops5 * 1.4 / us / 1e6:6.1f} on the 7 products executed); checksum {float(dq.float().abs().sum()):.6e}")
