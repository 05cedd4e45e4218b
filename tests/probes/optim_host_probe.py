#!/usr/bin/env python3
"""Host time of one optimizer step (clip + AdamW) over ~120 small tensors - what a launch-bound training step (5 clips) pays per optimizer:
HipAdamW against torch's clip_grad_norm_ + fused AdamW.  The tensors are tiny, so the wall time of the loop is the host's.  GPU box only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from titok_video_amd.optim import HipAdamW  # noqa: E402

DEV = torch.device("cuda:0")
N = int(os.environ.get("N", "120"))


def run(kind):
    ps = [torch.nn.Parameter(torch.randn(256 if i % 3 else 4096, device=DEV, dtype=torch.bfloat16)) for i in range(N)]
    opt = HipAdamW(ps, lr=1e-4) if kind == "hip" else torch.optim.AdamW(ps, lr=1e-4, fused=True)
    gs = [torch.randn_like(p) for p in ps]

    def step():
        for p, g in zip(ps, gs):
            p.grad = g
        if kind == "hip":
            opt.clip_and_step(1.0)
        else:
            torch.nn.utils.clip_grad_norm_(ps, 1.0)
            opt.step()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6


for kind in ("torch", "hip", "torch", "hip"):
    issue, wall = run(kind)
    print(f"{kind:6s} {N} tensors: host issue {issue:7.1f} us per step, wall {wall:7.1f} us per step")
