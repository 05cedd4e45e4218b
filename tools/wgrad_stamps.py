#!/usr/bin/env python3
"""Where the waves of k_wgrad128_bf16 spend their cycles (diagnostic build tools/wgrad_stamps.sh, loaded through TTV_LIB_PATH): s_memtime
sums per 64-token stage of the compute waves (barrier | sub-step 0 | sub-step 1) and of the loader waves (wait for the stage | barrier | issue).

    bash tools/wgrad_stamps.sh && TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_wgstamps.so python3 tools/wgrad_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd import _lib  # noqa: E402

DEV = torch.device("cuda:0")
lib = _lib.lib()
S = _lib.stream_ptr(DEV)
L = int(os.environ.get("L", "36864"))
NL = int(os.environ.get("NL", "4"))
for name, N, K in (("w3", 256, 704), ("w12", 1408, 256), ("out_proj", 256, 256), ("to_qkv", 768, 256)):
    dy = torch.randn(L, N, device=DEV).bfloat16()
    x = torch.randn(L, K, device=DEV).bfloat16()
    dw = torch.zeros(N, K, device=DEV)
    nb = int(lib.ttv_linear_wgrad_workspace_bytes(L, N, K))
    ws = torch.empty(max(nb, 4) // 4, device=DEV)
    blocks = nb // 65536
    n_s = (blocks + 36) // 37
    stamps = torch.zeros(n_s * (4 + NL) * 8, dtype=torch.int64, device=DEV)
    lib.ttv_debug_stamps(stamps.data_ptr())
    for _ in range(int(os.environ.get("REPS", "50"))):
        _lib.check(lib.ttv_linear_wgrad(dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), K, L, N, K, _lib.dtype_code(torch.bfloat16), ws.data_ptr(), nb, S), "wgrad")
    torch.cuda.synchronize()
    lib.ttv_debug_stamps(None)
    s = stamps.view(n_s, 4 + NL, 8).cpu().double()
    st = s[:, :, 5].clamp(min=1)
    seg = s[:, :, :3] / st[..., None]
    clk = s[:, :, :3].sum(-1) / s[:, :, 4].clamp(min=1) * 0.1
    c, l = seg[:, :4], seg[:, 4:]
    print(f"{name:9s} {blocks} blocks x {int(st[0, 0])} stages, clock {float(clk.mean()):.2f} GHz")
    print(f"   compute waves per stage: barrier {float(c[..., 0].mean()):6.0f}   sub-step 0 {float(c[..., 1].mean()):6.0f}   sub-step 1 {float(c[..., 2].mean()):6.0f}   total {float(c.sum(-1).mean()):6.0f} (min {float(c.sum(-1).min()):.0f}, max {float(c.sum(-1).max()):.0f})")
    print(f"   loader waves  per stage: data wait {float(l[..., 0].mean()):6.0f}   barrier {float(l[..., 1].mean()):6.0f}   issue {float(l[..., 2].mean()):6.0f}   total {float(l.sum(-1).mean()):6.0f}")
