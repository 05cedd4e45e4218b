#!/usr/bin/env python3
"""Is a kernel clock-limited by power?  The same launch on random and on all-zero activations (identical instruction stream, identical
cycle count; all-zero MFMA operands toggle nothing, so the part holds a higher clock: MI355X_MICROARCH.md 'DVFS give-back').  A large
ratio says the launch time on real data is set by the clock the part can afford, not by the kernel's cycles.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd import _lib  # noqa: E402
from titok_video_amd.plan import BatchPlan  # noqa: E402

lib = _lib.lib(); DEV = torch.device("cuda:0"); S = _lib.stream_ptr(DEV)
L, d, I = 36864, 256, 704
bf = torch.bfloat16


def t(fn, it=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


w12 = (torch.randn(2 * I, d, device=DEV) * d ** -0.5).to(bf); w3 = (torch.randn(d, I, device=DEV) * I ** -0.5).to(bf)
wo = (torch.randn(d, d, device=DEV) * d ** -0.5).to(bf)
gain = torch.ones(d, device=DEV); yb = torch.empty(L, d, dtype=bf, device=DEV)
mp = torch.empty(lib.ttv_mlp_pack_bytes(I, 0), dtype=torch.uint8, device=DEV)
lib.ttv_mlp_pack(w12.data_ptr(), w3.data_ptr(), wo.data_ptr(), None, 0, I, d, 0, mp.data_ptr(), S)
plan = BatchPlan([(16, 128, 128)] * 32, [128] * 32, (4, 8, 8), DEV)
table = plan.attention_table(4, 2)
for name, scale in (("random", 1.0), ("zero", 0.0)):
    x = (torch.randn(L, d, device=DEV) * scale).to(bf); ao = (torch.randn(L, d, device=DEV) * scale).to(bf)
    tail = lambda: lib.ttv_layer_tail_fused(ao.data_ptr(), d, gain.data_ptr(), 8.0, x.data_ptr(), d, mp.data_ptr(), I, yb.data_ptr(), d, gain.data_ptr(), 8.0, 1e-5, L, d, 0, None, S)
    print(f"{name:7s} activations: layer tail (k_mlp256<9>) {t(tail):7.1f} us", flush=True)
    qkv = (torch.randn(L, 768, device=DEV) * 0.9 * scale).to(bf)
    qkv[:, :256] = (qkv[:, :256].float() * (0.125 * 1.4426950408889634)).to(bf)
    out = torch.empty(L, 256, dtype=bf, device=DEV)
    for flags, nm in ((1 | 4 | 8, "k_attn_swp"), (1 | 4, "k_attn_bf16")):
        at = lambda: lib.ttv_attention(qkv.data_ptr(), 768, out.data_ptr(), 256, plan.cu_dev.data_ptr(), table.data_ptr(), table.shape[0], 4, 2, 64, flags, 0, S)
        print(f"{name:7s} activations: attention ({nm}) {t(at):7.1f} us", flush=True)
