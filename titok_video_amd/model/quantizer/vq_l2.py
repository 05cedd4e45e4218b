"""Nearest-codebook-entry (L2) vector quantiser on the HIP path (csrc/ttv_vq.hip).

Not a reference module: the reference quantises with FSQ only (model/quantizer/fsq.py).  BASELINE.json's north_star / configs
#4, #5 name this formulation ("nearest-codebook-entry L2 distance + straight-through lookup", codebooks 8192x32 / 16384x64), so it
is provided with FSQ's call shape:  `codes, {'indices': int32}` = vq(z).  `FSQ.lattice_codebook()` gives the codebook on which this
quantiser reproduces FSQ's own indices (applied to `FSQ.bounded(z)`).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import _lib


class L2Quantizer(nn.Module):
    """codebook [N, C] (a Parameter: it can be trained by whatever codebook loss the caller adds; none is implied here).

    forward(z [rows, C]) -> (codes [rows, C] with straight-through gradient d codes / d z = I, {'indices': int32 [rows]})."""

    def __init__(self, codebook: torch.Tensor):
        super().__init__()
        if codebook.dim() != 2 or codebook.shape[1] > 64:
            raise ValueError("codebook must be [N, C] with C <= 64")
        self.codebook = nn.Parameter(codebook.detach().clone())
        self._norms = None
        self._norms_key = None

    @property
    def codebook_size(self) -> int:
        return int(self.codebook.shape[0])

    @property
    def codebook_dim(self) -> int:
        return int(self.codebook.shape[1])

    def _cb(self, dtype):
        cb = self.codebook.detach()
        key = (cb.data_ptr(), self.codebook._version, dtype, str(cb.device))
        if self._norms_key != key:
            cbd = cb.to(dtype).contiguous()
            norms = torch.empty(cbd.shape[0], dtype=torch.float32, device=cbd.device)
            _lib.check(_lib.lib().ttv_vq_codebook_norms(cbd.data_ptr(), _lib.dtype_code(dtype), cbd.shape[1], cbd.shape[0], cbd.shape[1],
                                                        norms.data_ptr(), _lib.stream_ptr(cbd.device)), "ttv_vq_codebook_norms")
            self._norms, self._norms_key = (cbd, norms), key
        return self._norms

    @torch.no_grad()
    def indices(self, z: torch.Tensor, want_distance: bool = False):
        _lib.require_gpu(z, "L2Quantizer")
        if z.dim() != 2 or z.shape[1] != self.codebook_dim:
            raise ValueError(f"z must be [rows, {self.codebook_dim}]")
        z = z.contiguous()
        cbd, norms = self._cb(z.dtype)
        idx = torch.empty(z.shape[0], dtype=torch.int32, device=z.device)
        dist = torch.empty(z.shape[0], dtype=torch.float32, device=z.device) if want_distance else None
        ws = torch.empty(int(_lib.lib().ttv_vq_workspace_bytes(z.shape[0])) // 8, dtype=torch.int64, device=z.device)
        _lib.check(_lib.lib().ttv_vq_l2_argmin(z.data_ptr(), _lib.dtype_code(z.dtype), z.shape[1], cbd.data_ptr(), cbd.shape[1], norms.data_ptr(),
                                               z.shape[0], cbd.shape[0], cbd.shape[1], idx.data_ptr(), _lib.ptr(dist), ws.data_ptr(),
                                               ws.numel() * 8, _lib.stream_ptr(z.device)), "ttv_vq_l2_argmin")
        return (idx, dist) if want_distance else idx

    @torch.no_grad()
    def lookup(self, indices: torch.Tensor, dtype=None) -> torch.Tensor:
        dtype = dtype or self.codebook.dtype
        cbd, _ = self._cb(dtype)
        idx = indices.to(torch.int32).contiguous()
        out = torch.empty((idx.shape[0], cbd.shape[1]), dtype=dtype, device=idx.device)
        _lib.check(_lib.lib().ttv_vq_lookup(cbd.data_ptr(), _lib.dtype_code(dtype), cbd.shape[1], idx.data_ptr(), idx.shape[0], cbd.shape[1],
                                            out.data_ptr(), cbd.shape[1], _lib.stream_ptr(idx.device)), "ttv_vq_lookup")
        return out

    def forward(self, z: torch.Tensor):
        """Straight-through quantiser: VALUE codebook[idx]; gradient identity towards z (the reference's FSQ estimator, fsq.py:48-51)
        and, through the lookup, towards the codebook rows that were selected (scatter-add of the decoder's token gradient:
        `_LookupFn`) - the one quantiser slot of the reference is trained end to end by the reconstruction loss (titok.py:37,47-52,
        train.py:65-83); any commitment / codebook loss is the caller's to add."""
        idx = self.indices(z.detach())
        if torch.is_grad_enabled() and (z.requires_grad or self.codebook.requires_grad):
            q = _LookupFn.apply(self, idx, z.dtype, self.codebook)
            codes = q + (z - z.detach()) if z.requires_grad else q
        else:
            codes = self.lookup(idx, z.dtype)
        return codes, {"indices": idx}


class _LookupFn(torch.autograd.Function):
    """codes = codebook[idx] on the HIP path; backward = ttv_vq_lookup_backward (fp32 scatter-add into the codebook gradient)."""

    @staticmethod
    def forward(ctx, vq: "L2Quantizer", idx: torch.Tensor, dtype, codebook: torch.Tensor):
        ctx.idx, ctx.shape, ctx.cb_dtype = idx, tuple(codebook.shape), codebook.dtype
        return vq.lookup(idx, dtype)

    @staticmethod
    def backward(ctx, dcodes):
        dcodes = dcodes.contiguous()
        if dcodes.dtype not in (torch.float32, torch.bfloat16):
            dcodes = dcodes.float()
        dcb = torch.zeros(ctx.shape, dtype=torch.float32, device=dcodes.device)
        _lib.check(_lib.lib().ttv_vq_lookup_backward(dcodes.data_ptr(), _lib.dtype_code(dcodes.dtype), dcodes.shape[1], ctx.idx.data_ptr(), dcodes.shape[0],
                                                     dcodes.shape[1], dcb.data_ptr(), ctx.shape[1], _lib.stream_ptr(dcodes.device)), "ttv_vq_lookup_backward")
        return None, None, None, dcb.to(ctx.cb_dtype)
