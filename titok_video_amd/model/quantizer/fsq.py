"""FSQ with the reference's API (model/quantizer/fsq.py:55-135) on the HIP path.

`forward` (bound -> round -> normalise -> mixed-radix index) and `indices_to_codes` run in csrc/ttv_elem.hip.
The per-channel constants (half_l, offset, shift) are evaluated once on the host with the same fp32 tensor
expressions as the reference (fsq.py:80-82) and handed to the kernels, so both sides round them identically.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.nn as nn

from ... import _lib


class FSQ(nn.Module):
    def __init__(self, levels: List[int], dim: Optional[int] = None):
        super().__init__()
        levels = [int(l) for l in levels]
        if not 1 <= len(levels) <= _lib.TTV_MAX_FSQ:
            raise ValueError(f"FSQ supports 1..{_lib.TTV_MAX_FSQ} levels")
        _levels = torch.tensor(levels, dtype=torch.int32)
        self.register_buffer("_levels", _levels, persistent=False)
        _basis = torch.cumprod(torch.tensor([1] + levels[:-1]), dim=0, dtype=torch.int32)
        self.register_buffer("_basis", _basis, persistent=False)
        self.codebook_dim = len(levels)
        self.dim = dim if dim is not None else len(levels)
        self.codebook_size = int(_levels.prod().item())

        eps = 1e-3
        half_l = (_levels - 1) * (1 + eps) / 2                      # fsq.py:80 (fp32)
        offset = torch.where(_levels % 2 == 0, 0.5, 0.0)            # fsq.py:81
        shift = (offset / half_l).atanh()                           # fsq.py:82
        half_width = (_levels // 2).to(torch.float32)               # fsq.py:89
        p = _lib.FsqParams()
        p.n = len(levels)
        for i in range(len(levels)):
            p.levels[i], p.basis[i] = levels[i], int(_basis[i])
            p.half_l[i], p.offset[i] = float(half_l[i]), float(offset[i])
            p.shift[i], p.half_width[i] = float(shift[i]), float(half_width[i])
        self.params = p

        # implicit_codebook = indices_to_codes(arange(size)) (fsq.py:75): exact small-integer arithmetic on the host
        idx = torch.arange(self.codebook_size, dtype=torch.int32).unsqueeze(-1)
        lvl = (idx // _basis) % _levels
        hw = _levels // 2
        self.register_buffer("implicit_codebook", (lvl - hw) / hw, persistent=False)

    def lattice_codebook(self) -> torch.Tensor:
        """fp32 [codebook_size, C]: the implicit codebook in LATTICE units, implicit_codebook * (levels // 2) (fsq.py:73-76, 89).
        The entry nearest (L2) to `bounded(z)` is FSQ's own index away from rounding ties - the codebook on which
        quantizer.vq_l2.L2Quantizer reproduces this module."""
        return (self.implicit_codebook * (self._levels // 2).to(torch.float32)).contiguous()

    # -- hot path ---------------------------------------------------------------------------------
    def _run(self, z: torch.Tensor, want_bounded: bool):
        _lib.require_gpu(z, "FSQ.forward")
        if z.dim() != 2 or z.shape[1] != self.codebook_dim:
            raise ValueError(f"FSQ expects [rows, {self.codebook_dim}], got {tuple(z.shape)}")
        z = z.contiguous()
        code = _lib.dtype_code(z.dtype)
        rows = z.shape[0]
        codes = torch.empty_like(z)
        indices = torch.empty((rows,), dtype=torch.int32, device=z.device)
        b = torch.empty((rows, self.codebook_dim), dtype=torch.float32, device=z.device) if want_bounded else None
        rc = _lib.lib().ttv_fsq_forward(C.byref(self.params), z.data_ptr(), code, rows, codes.data_ptr(), code,
                                        indices.data_ptr(), _lib.ptr(b), _lib.stream_ptr(z.device))
        _lib.check(rc, "ttv_fsq_forward")
        return codes, indices, b

    def forward(self, z: torch.Tensor):
        """z [rows, C] -> (codes in z.dtype, {'indices': int32 [rows]}) - fsq.py:123-135.  With grad enabled and a
        differentiable z the rounding is straight-through (fsq.py:48-51), backward in csrc/ttv_bwd.hip."""
        if torch.is_grad_enabled() and z.requires_grad:
            codes, indices = _FsqFn.apply(z, self)
            return codes, {"indices": indices}
        codes, indices, _ = self._run(z, False)
        return codes, {"indices": indices}

    def bounded(self, z: torch.Tensor) -> torch.Tensor:
        """fp32 `bound(z)` (value before rounding) from the same kernel; used for rounding-margin reports."""
        return self._run(z, True)[2]

    def indices_to_codes(self, indices: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """int32 [rows] -> codes [rows, C] - fsq.py:100-121."""
        assert indices is not None
        _lib.require_gpu(indices, "FSQ.indices_to_codes")
        shape = tuple(indices.shape)
        flat = indices.reshape(-1).to(torch.int32).contiguous()
        codes = torch.empty((flat.shape[0], self.codebook_dim), dtype=dtype, device=indices.device)
        rc = _lib.lib().ttv_fsq_indices_to_codes(C.byref(self.params), flat.data_ptr(), flat.shape[0], codes.data_ptr(),
                                                 _lib.dtype_code(dtype), _lib.stream_ptr(indices.device))
        _lib.check(rc, "ttv_fsq_indices_to_codes")
        return codes.reshape(*shape, self.codebook_dim)

    # -- API helpers that are not on the hot path (tensor expressions, any device) -----------------
    def bound(self, z, eps: float = 1e-3):
        lv = self._levels.to(z.device)
        half_l = (lv - 1) * (1 + eps) / 2
        offset = torch.where(lv % 2 == 0, 0.5, 0.0)
        shift = (offset / half_l).atanh()
        return (z + shift).tanh() * half_l - offset

    def codes_to_indices(self, zhat):
        hw = (self._levels // 2).to(zhat.device)
        return (((zhat * hw) + hw) * self._basis.to(zhat.device)).sum(dim=-1).to(torch.int32)

    def indices_to_level_indices(self, indices):
        return (indices.unsqueeze(-1) // self._basis.to(indices.device)) % self._levels.to(indices.device)


class _FsqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, fsq):
        codes, indices, _ = fsq._run(z, False)
        ctx.fsq = fsq
        ctx.save_for_backward(z)
        ctx.mark_non_differentiable(indices)
        return codes, indices

    @staticmethod
    def backward(ctx, dcodes, _dindices):
        (z,) = ctx.saved_tensors
        fsq = ctx.fsq
        zf = z.detach().float().contiguous()
        dcodes = dcodes.contiguous()
        dz = torch.empty_like(zf)
        rc = _lib.lib().ttv_fsq_backward(C.byref(fsq.params), zf.data_ptr(), dcodes.data_ptr(), _lib.dtype_code(dcodes.dtype),
                                         dz.data_ptr(), zf.shape[0], _lib.stream_ptr(z.device))
        _lib.check(rc, "ttv_fsq_backward")
        return dz.to(z.dtype), None
