#!/usr/bin/env python3
"""The general-K bf16 GEMM at the base tower's shapes (L = 36 864 rows, d = 768): the 128 x 128-tile kernel (k_gemm_bf16_dma) against the
256 x 256-tile kernel (k_gemm_bf16_t256), per epilogue.  GPU box only.   python tools/gemm_t256_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd import _lib  # noqa: E402
from titok_video_amd.plan import BatchPlan  # noqa: E402

DEV = torch.device("cuda:0")
lib = _lib.lib()
ST = _lib.stream_ptr(DEV)
bf = torch.bfloat16
code = _lib.dtype_code(bf)
B = int(os.environ.get("B", "4"))
plan = BatchPlan([(32, 256, 256)] * B, [1024] * B, (4, 8, 8), DEV)
M, d, gq, I = plan.total_rows, 768, 256, 2048
g = torch.Generator().manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(bf).to(DEV)


x768, x2048 = rnd(M, d), rnd(M, I)
cases = {
    "to_qkv + rotary   N=2048 K=768 ": (2 * M * (2 * d + 2 * gq) * d, lambda w=rnd(2 * d + 2 * gq, d, scale=d ** -0.5), y=torch.empty(M, 2 * d + 2 * gq, dtype=bf, device=DEV):
                                        lib.ttv_linear_qkv_rope(x768.data_ptr(), d, w.data_ptr(), d, y.data_ptr(), 2 * d + 2 * gq, M, d, gq, plan.rope_cs.data_ptr(), code, ST)),
    "w12 + GEGLU       N=2x2048 K=768": (2 * M * 2 * I * d, lambda w=rnd(2 * I, d, scale=d ** -0.5), y=torch.empty(M, I, dtype=bf, device=DEV):
                                        lib.ttv_linear_geglu(x768.data_ptr(), d, w.data_ptr(), d, y.data_ptr(), I, M, I, d, code, ST)),
    "out_proj + resid  N=768 K=768  ": (2 * M * d * d, lambda w=rnd(d, d, scale=d ** -0.5), r=rnd(M, d), y=torch.empty(M, d, dtype=torch.float32, device=DEV):
                                        lib.ttv_linear_residual(x768.data_ptr(), d, w.data_ptr(), d, r.data_ptr(), d, 8.0, y.data_ptr(), d, 1, M, d, d, code, ST)),
    "w3 + resid        N=768 K=2048 ": (2 * M * d * I, lambda w=rnd(d, I, scale=I ** -0.5), r=rnd(M, d), y=torch.empty(M, d, dtype=torch.float32, device=DEV):
                                        lib.ttv_linear_residual(x2048.data_ptr(), I, w.data_ptr(), I, r.data_ptr(), d, 8.0, y.data_ptr(), d, 1, M, d, I, code, ST)),
    "plain store       N=2048 K=768 ": (2 * M * 2048 * d, lambda w=rnd(2048, d, scale=d ** -0.5), y=torch.empty(M, 2048, dtype=bf, device=DEV):
                                        lib.ttv_linear(x768.data_ptr(), d, w.data_ptr(), d, None, None, y.data_ptr(), 2048, M, 2048, d, code, ST)),
}
print(f"{M} rows (batch {B} x 9216)")
for name, (flops, call) in cases.items():
    row = []
    nostore = 1 if os.environ.get("NOSTORE") == "1" else 0      # ttv_debug_set bit 0: the epilogue keeps its values alive and stores nothing
    for bit, tag in ((1024, "128x128"), (512, "256x256")):
        lib.ttv_debug_set(bit | nostore)
        for _ in range(3):
            _lib.check(call(), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        row.append(f"{tag} {us:7.1f} us {flops / us / 1e6:6.0f} TFLOP/s = {flops / us / 1e6 / 2500:.3f}")
    lib.ttv_debug_set(0)
    print(f"{name}  " + "   |   ".join(row))

if os.environ.get("CHECK", "1") == "1":
    # race screen of the 256 x 256 schedule: the same launch 200 times at two shapes, every result compared bit for bit with the
    # 128 x 128 kernel's (same products, same k order)
    for (N, K) in ((2048, 768), (768, 2048)):
        x = rnd(M, K)
        w = rnd(N, K, scale=K ** -0.5)
        y0 = torch.empty(M, N, dtype=bf, device=DEV)
        y1 = torch.empty(M, N, dtype=bf, device=DEV)
        lib.ttv_debug_set(1024)
        _lib.check(lib.ttv_linear(x.data_ptr(), K, w.data_ptr(), K, None, None, y0.data_ptr(), N, M, N, K, code, ST), "linear")
        lib.ttv_debug_set(512)
        bad = 0
        for it in range(200):
            y1.fill_(float("nan"))
            _lib.check(lib.ttv_linear(x.data_ptr(), K, w.data_ptr(), K, None, None, y1.data_ptr(), N, M, N, K, code, ST), "linear")
            if not torch.equal(y0, y1):
                bad += 1
        lib.ttv_debug_set(0)
        print(f"race screen N={N} K={K}: {bad} of 200 launches differ from the 128 x 128 kernel")
