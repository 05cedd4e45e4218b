#!/usr/bin/env python3
"""Per-kernel timing of the C-ABI ops at the benchmark shapes (L = 36864 rows, tiny dims).  GPU box only."""
import ctypes as C
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
L, d, g, I = plan.total_rows, 256, 128, 704
bf = torch.bfloat16


def rnd(*shape, dt=bf, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).to(dt)


def timeit(name, fn, flops=None, bytes_=None, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    extra = ""
    if flops:
        extra += f"  {flops / us / 1e6:8.1f} TFLOP/s"
    if bytes_:
        extra += f"  {bytes_ / us / 1e6:6.2f} TB/s"
    print(f"{name:44s} {us:8.1f} us{extra}", flush=True)


x = rnd(L, d)
wqkv = rnd(2 * d + 2 * g, d, scale=d ** -0.5)
w12 = rnd(2 * I, d, scale=d ** -0.5)
wo = rnd(d, d, scale=d ** -0.5)
w3 = rnd(d, I, scale=I ** -0.5)
wpo = rnd(768, d, scale=d ** -0.5)
wpi = rnd(d, 768, scale=768 ** -0.5)
qkv = torch.empty(L, 2 * d + 2 * g, dtype=bf, device=DEV)
h = rnd(L, I)
y32 = torch.empty(L, d, dtype=torch.float32, device=DEV)
yb = torch.empty(L, d, dtype=bf, device=DEV)
hb = torch.empty(L, I, dtype=bf, device=DEV)
big = torch.empty(32768, 768, dtype=bf, device=DEV)
pat = rnd(32768, 768)
gain = torch.ones(d, device=DEV)
ao = torch.empty(L, d, dtype=bf, device=DEV)
qkv_in = rnd(L, 2 * d + 2 * g)

QUICK = os.environ.get("QUICK")
for dbg in ([0, 1] if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]):
    lib.ttv_debug_set(dbg)
    print(f"---- debug flags = {dbg} ----")
    timeit("linear K256 N768 (k256, store)", lambda: lib.ttv_linear(x.data_ptr(), d, wqkv.data_ptr(), d, None, None, qkv.data_ptr(), 768, L, 768, d, 0, S),
           2.0 * L * d * 768, L * (d + 768) * 2)
    timeit("qkv_rope K256 N768 (k256)", lambda: lib.ttv_linear_qkv_rope(x.data_ptr(), d, wqkv.data_ptr(), d, qkv.data_ptr(), 768, L, d, g, plan.rope_cs.data_ptr(), 0, S),
           2.0 * L * d * 768, L * (d + 768) * 2)
    timeit("geglu K256 I704 (k256)", lambda: lib.ttv_linear_geglu(x.data_ptr(), d, w12.data_ptr(), d, hb.data_ptr(), I, L, I, d, 0, S),
           2.0 * L * d * 2 * I, L * (d + I) * 2)
    timeit("residual f32-out K256 N256 (k256)", lambda: lib.ttv_linear_residual(x.data_ptr(), d, wo.data_ptr(), d, x.data_ptr(), d, 8.0, y32.data_ptr(), d, 1, L, d, d, 0, S),
           2.0 * L * d * d, L * d * (2 + 2 + 4))
    timeit("residual+norm fused K256 N256 (rownorm)", lambda: lib.ttv_linear_residual_norm(x.data_ptr(), d, wo.data_ptr(), d, yb.data_ptr(), d, 8.0, gain.data_ptr(), 1e-5, yb.data_ptr(), d, L, d, d, 0, S),
           2.0 * L * d * d, L * d * 6)
    mpack = torch.empty(lib.ttv_mlp_pack_bytes(I, 768), dtype=torch.uint8, device=DEV)
    lib.ttv_mlp_pack(w12.data_ptr(), w3.data_ptr(), wo.data_ptr(), wqkv.data_ptr(), 768, I, d, 0, mpack.data_ptr(), S)
    timeit("fused MLP (w12+geglu+w3+keel+norm)", lambda: lib.ttv_mlp_fused(x.data_ptr(), d, mpack.data_ptr(), I, yb.data_ptr(), d, gain.data_ptr(), 8.0, 1e-5, L, d, 0, S),
           2.0 * L * d * 3 * I, L * d * 4)
    timeit("layer tail fused (out_proj+keel+mlp+keel)", lambda: lib.ttv_layer_tail_fused(ao.data_ptr(), d, gain.data_ptr(), 8.0, x.data_ptr(), d, mpack.data_ptr(), I, yb.data_ptr(), d, gain.data_ptr(), 8.0, 1e-5, L, d, 0, None, S),
           2.0 * L * d * (3 * I + d), L * d * 6)
    nxq = _lib.NextQkv(qkv=qkv.data_ptr(), ld=768, rope_cs=plan.rope_cs.data_ptr(), rows=768, rope_q_end=256, rope_k_begin=512, rope_k_end=640)
    timeit("layer tail fused + next qkv/rope", lambda: lib.ttv_layer_tail_fused(ao.data_ptr(), d, gain.data_ptr(), 8.0, x.data_ptr(), d, mpack.data_ptr(), I, yb.data_ptr(), d, gain.data_ptr(), 8.0, 1e-5, L, d, 0, C.byref(nxq), S),
           2.0 * L * d * (3 * I + d + 768), L * d * 6 + L * 768 * 2)
    timeit("residual+norm fused K704 N256 (rowtile)", lambda: lib.ttv_linear_residual_norm(h.data_ptr(), I, w3.data_ptr(), I, yb.data_ptr(), d, 8.0, gain.data_ptr(), 1e-5, yb.data_ptr(), d, L, d, I, 0, S),
           2.0 * L * I * d, L * (I * 2 + d * 4))
    if QUICK:
        continue
    timeit("residual bf16-out K256 N256 (k256)", lambda: lib.ttv_linear_residual(x.data_ptr(), d, wo.data_ptr(), d, x.data_ptr(), d, 1.0, yb.data_ptr(), d, 0, L, d, d, 0, S),
           2.0 * L * d * d, L * d * 6)
    timeit("residual f32-out K704 N256 (generic)", lambda: lib.ttv_linear_residual(h.data_ptr(), I, w3.data_ptr(), I, x.data_ptr(), d, 8.0, y32.data_ptr(), d, 1, L, d, I, 0, S),
           2.0 * L * I * d, L * (I * 2 + d * 6))
    timeit("linear K768 N256 +bias (generic, proj_in)", lambda: lib.ttv_linear(pat.data_ptr(), 768, wpi.data_ptr(), 768, None, None, yb.data_ptr(), d, 32768, d, 768, 0, S),
           2.0 * 32768 * 768 * d, 32768 * (768 + d) * 2)
    timeit("linear K256 N768 (k256, dec proj_out)", lambda: lib.ttv_linear(x.data_ptr(), d, wpo.data_ptr(), d, None, None, big.data_ptr(), 768, 32768, 768, d, 0, S),
           2.0 * 32768 * 768 * d, 32768 * (768 + d) * 2)
lib.ttv_debug_set(0)
timeit("attention (gate)", lambda: lib.ttv_attention(qkv_in.data_ptr(), 768, ao.data_ptr(), d, plan.cu_dev.data_ptr(), plan.attention_table(4, 2).data_ptr(), plan.attention_table(4, 2).shape[0], 4, 2, 64, 1, 0, S),
       32 * 4.0 * 1152 * 1152 * d)
timeit("rmsnorm bf16->bf16", lambda: lib.ttv_rmsnorm(x.data_ptr(), 0, d, None, yb.data_ptr(), 0, d, None, gain.data_ptr(), L, d, 1e-5, S), None, L * d * 4)
timeit("rmsnorm f32->bf16", lambda: lib.ttv_rmsnorm(y32.data_ptr(), 1, d, None, yb.data_ptr(), 0, d, None, gain.data_ptr(), L, d, 1e-5, S), None, L * d * 6)
