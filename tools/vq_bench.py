#!/usr/bin/env python3
"""L2-argmin quantiser at the codebook sizes of BASELINE configs #4 / #5: time, TFLOP/s (2 * rows * N * C) against the MFMA peak of
the input dtype.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd.model.quantizer.vq_l2 import L2Quantizer  # noqa: E402

DEV = "cuda:0"
for N, C, rows in ((8192, 32, 32768), (16384, 64, 32768), (4375, 5, 4096)):
    for dtype, peak in ((torch.bfloat16, 2500.0), (torch.float32, 157.3)):
        vq = L2Quantizer(torch.randn(N, C)).to(DEV)
        z = torch.randn(rows, C, device=DEV).to(dtype)
        for _ in range(3):
            vq.indices(z)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            vq.indices(z)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        fl = 2.0 * rows * N * C
        print(f"N {N:6d} C {C:3d} rows {rows:6d} {str(dtype):15s} {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s = {fl / us / 1e6 / peak:.3f} of peak", flush=True)
