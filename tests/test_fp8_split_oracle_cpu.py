"""CPU checks of the round-4 test infrastructure (no GPU): the OCP-MX e4m3 oracle (oracle/fp8_oracle.py) and the split-bf16 emulation
the GPU tests' tolerances come from (tests/probes/split_bf16_probe.py).  Neither has a reference counterpart (the reference runs bf16
autocast only): these tests pin the restated definitions against their own published properties."""
import os
import sys

import torch

from oracle import fp8_oracle as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "probes"))


def test_mx_quantize_properties():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 256, generator=g) * torch.exp(4 * torch.randn(64, 8, generator=g)).repeat_interleave(32, 1)
    x[3] = 0
    x[5, 32:64] = 0
    for row_scaled in (False, True):
        q, e, rs = F.mx_quantize(x, row_scaled)
        assert q.dtype == torch.uint8 and q.shape == x.shape and e.shape == (64, 8)
        v = q.view(torch.float8_e4m3fn).float()
        assert float(v.abs().max()) <= 448.0                                    # no saturation: the scale brings every block into range
        y = x / rs[:, None] if row_scaled else x
        blk = y.view(64, 8, 32).abs().amax(2)
        scale = torch.ldexp(torch.ones(64, 8), e.to(torch.int32) - 127)
        nz = blk > 0
        assert bool(((blk / scale)[nz] <= 448.0 * (1 + 1e-6)).all())             # 2^E >= max / 448 ...
        assert bool(((blk / scale)[nz] > 224.0 * (1 - 1e-6)).all())              # ... and the SMALLEST such power of two
        assert bool((e[~nz] == 127).all())                                      # all-zero blocks: scale 1
        d = F.mx_dequantize(q, e, rs)
        floor = (scale * 2.0 ** -6).repeat_interleave(32, 1).double() * (rs.double()[:, None] if row_scaled else 1.0)
        assert float(((d - x.double()).abs() / (x.double().abs() + floor)).max()) <= 2 ** -4 + 1e-6     # half an e4m3 ulp in the normal range
        if row_scaled:
            assert torch.allclose(rs[rs != 1], (x.abs().amax(1) / 448.0)[rs != 1], rtol=1e-6)


def test_mx_scale_layout_is_a_permutation_of_the_blocks():
    e = torch.arange(3 * 24, dtype=torch.uint8).view(3, 24)          # K = 768: 24 blocks, 6 k-tiles, padded to 8
    lay = F.mx_scale_layout(e)
    assert lay.shape == (3, 32)
    for b in range(24):
        assert bool((lay[:, (b & 3) * 8 + (b >> 2)] == e[:, b]).all())
    assert int((lay != 0).sum()) == int((e != 0).sum())                # the padding bytes stay zero


def test_split_bf16_products_error_ladder():
    """ah bh alone (plain bf16 operands) ~ 2^-9, three passes ~ 2^-17, six passes (three-term split) ~ fp32: the ladder the GPU
    tolerances of tests/test_hip_split3.py are taken from."""
    import split_bf16_probe as P
    g = torch.Generator().manual_seed(1)
    a = torch.randn(96, 256, generator=g)
    b = torch.randn(256, 80, generator=g)
    ref = a.double() @ b.double()
    rel = lambda y: float((y.double() - ref).norm() / ref.norm())
    e1, e3, e4, e6 = (rel(P.mm(a, b, p)) for p in (1, 3, 4, 6))
    assert 1e-3 < e1 < 5e-3
    assert e3 < 1.5e-5 and e4 <= e3 * 1.05
    assert e6 < 5e-7
    hi, lo = P.split(a, 2)
    assert torch.equal(hi, a.to(torch.bfloat16).float())
    assert float((a - hi - lo).abs().max() / a.abs().max()) < 2 ** -16
