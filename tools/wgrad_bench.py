#!/usr/bin/env python3
"""Weight-gradient GEMM timing at the training-step shapes (L = 36864 rows, tiny dims).  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd import _lib  # noqa: E402

DEV = torch.device("cuda:0")
lib = _lib.lib()
S = _lib.stream_ptr(DEV)
L = int(os.environ.get("L", "36864"))
for name, N, K in (("w3", 256, 704), ("w12", 1408, 256), ("out_proj", 256, 256), ("to_qkv", 768, 256), ("proj_in", 256, 768)):
    dy = torch.randn(L, N, device=DEV).bfloat16()
    x = torch.randn(L, K, device=DEV).bfloat16()
    dw = torch.zeros(N, K, device=DEV)
    nb = int(lib.ttv_linear_wgrad_workspace_bytes(L, N, K)) if os.environ.get("WS", "1") == "1" else 0
    ws = torch.empty(max(nb, 4) // 4, device=DEV)
    fn = lambda: _lib.check(lib.ttv_linear_wgrad(dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), K, L, N, K, _lib.dtype_code(torch.bfloat16),
                                                 ws.data_ptr() if nb else None, nb, S), "wgrad")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"wgrad {name:9s} N={N:5d} K={K:4d}  {us:7.1f} us  {2.0 * L * N * K / us / 1e6:7.1f} TFLOP/s")

# the dX GEMMs of the same layers (general-K kernel, 256 output features): grid balance decides the token-tile height
for name, N, K in (("dX qkv", 256, 768), ("dX w12", 256, 1408), ("fwd w3", 256, 704)):
    x = torch.randn(L, K, device=DEV).bfloat16()
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
    y = torch.empty(L, N, device=DEV, dtype=torch.bfloat16)
    for flag, tag in ((256, "tile 128"), (128, "tile 160"), (0, "auto")):
        lib.ttv_debug_set(flag)
        fn = lambda: _lib.check(lib.ttv_linear(x.data_ptr(), K, w.data_ptr(), K, None, None, y.data_ptr(), N, L, N, K, _lib.dtype_code(torch.bfloat16), S), "linear")
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{name:8s} N={N:4d} K={K:4d} {tag:9s} {us:7.1f} us  {2.0 * L * N * K / us / 1e6:7.1f} TFLOP/s")
    lib.ttv_debug_set(0)
