"""CPU oracle for the nearest-codebook-entry (L2) quantiser.

TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).  NOT REFERENCE-PINNED: the reference has no learned-codebook
quantiser - its only quantiser is FSQ (model/quantizer/fsq.py:78-135).  BASELINE.json's north_star and configs #4 / #5 ask for the
L2 formulation, so this states the textbook definition (cdist + argmin, lowest index on ties - torch.argmin's rule) in float64;
the tie to the reference is the lattice property tested in tests/test_hip_vq.py: on  implicit_codebook * (levels // 2)
(fsq.py:73-76) applied to FSQ.bound(z) (fsq.py:78-83) the nearest entry IS FSQ's index (fsq.py:105-109) away from rounding ties.
"""
from __future__ import annotations

import torch


def l2_argmin(z: torch.Tensor, codebook: torch.Tensor, chunk: int = 4096):
    """(int32 indices [rows], float64 best squared distance [rows], float64 gap to the runner-up [rows])."""
    zd, cd = z.double(), codebook.double()
    idx, best, gap = [], [], []
    for i in range(0, zd.shape[0], chunk):
        d = torch.cdist(zd[i:i + chunk], cd, p=2).pow(2)           # [rows, N]
        two = torch.topk(d, k=min(2, d.shape[1]), dim=1, largest=False)
        idx.append(torch.argmin(d, dim=1))                          # first minimal index
        best.append(two.values[:, 0])
        gap.append(two.values[:, 1] - two.values[:, 0] if d.shape[1] > 1 else torch.full_like(two.values[:, 0], float("inf")))
    return torch.cat(idx).to(torch.int32), torch.cat(best), torch.cat(gap)


def fsq_lattice(levels):
    """FSQ's implicit codebook in lattice units: entry n = (digits of n in the mixed radix `levels`) - levels // 2 (fsq.py:73-76,
    100-121: indices_to_codes * half_width)."""
    lv = torch.tensor(list(levels), dtype=torch.int64)
    basis = torch.cumprod(torch.tensor([1] + list(levels[:-1]), dtype=torch.int64), dim=0)
    n = int(torch.prod(lv))
    idx = torch.arange(n, dtype=torch.int64)
    digits = (idx[:, None] // basis[None, :]) % lv[None, :]
    return (digits - (lv // 2)[None, :]).to(torch.float32)
