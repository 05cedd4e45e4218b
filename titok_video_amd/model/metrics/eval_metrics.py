"""PSNR of the evaluation loop on the GPU (reference model/metrics/eval_metrics.py:11-51, the 'psnr' entry).

The reference builds torchmetrics' PeakSignalNoiseRatio(data_range=2) and feeds it `x.clamp(-1, 1)` per clip (eval_metrics.py:19,
32-36): a running sum of squared errors and an element count, `10 * log10(data_range^2 / mse)` at compute().  Here both sums live in
one device buffer filled by `ttv_sq_err_accumulate` (one launch per update, no host sync until compute()).  SSIM / FVD / JEDi are
out of scope (remote weights).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Sequence

import torch
import torch.nn as nn

from ... import _lib


class EvalMetrics(nn.Module):
    def __init__(self, config=None, eval_prefix: str = "eval"):
        super().__init__()
        self.eval_prefix = eval_prefix
        names = ["psnr"]
        if config is not None:
            names = [m for m in config.training.eval.log_metrics]
            for m in names:
                if m != "psnr":
                    raise NotImplementedError(f"metric '{m}' needs weights fetched over the network (reference model/metrics/); only 'psnr' is built")
        self.names = names
        self._acc = None

    def update(self, recon: Sequence[torch.Tensor], target: Sequence[torch.Tensor]) -> None:
        if not self.names:
            return
        r0 = recon[0]
        _lib.require_gpu(r0, "EvalMetrics.update")
        if self._acc is None or self._acc.device != r0.device:
            self._acc = torch.zeros(2, dtype=torch.float64, device=r0.device)
        rs = [t.contiguous() for t in recon]
        ts = [t.to(r0.dtype).contiguous() for t in target]
        sizes = (C.c_int32 * len(rs))(*[int(t.numel()) for t in rs])
        rc = _lib.lib().ttv_sq_err_accumulate(_lib.ptr_array(rs), _lib.ptr_array(ts), sizes, len(rs), _lib.dtype_code(r0.dtype), 1,
                                              self._acc.data_ptr(), _lib.stream_ptr(r0.device))
        _lib.check(rc, "ttv_sq_err_accumulate")

    def compute(self) -> dict:
        if not self.names or self._acc is None:
            return {}
        sq, n = (float(v) for v in self._acc.cpu())
        psnr = 10.0 * math.log10(4.0 * n / sq) if sq > 0 else float("inf")
        return {f"{self.eval_prefix}/psnr": psnr}

    def reset(self) -> None:
        if self._acc is not None:
            self._acc.zero_()
