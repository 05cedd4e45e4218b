"""TEST INFRASTRUCTURE - CPU restatement of the block-scaled (OCP MX) e4m3 quantisation of BASELINE.json configs[4] ("mixed bf16/fp8
MFMA").  Only tests/, __graft_entry__.smoke() and bench.py's checking legs may import this; the product path never does.

PARITY UNPINNED against the reference: the reference has no fp8 path (it trains under Lightning bf16-mixed autocast,
configs/tiny.yaml:70), so there is no reference code or fixture for this arithmetic.  What is restated here is the published OCP
Microscaling Formats (MX) v1.0 definition - MXFP8 with E4M3 elements: 32 consecutive elements share one E8M0 scale X = 2^E; an element
is stored as round_to_nearest_even_e4m3(v / X) - with the scale rule this library uses, E = ceil(log2(max|block| / 448)): the smallest
power of two that brings the block into the e4m3 range (|q| <= 448; no saturation, no extra headroom).  e4m3 rounding itself is
torch.float8_e4m3fn (OCP e4m3, round to nearest even)."""
import math

import torch

E4M3_MAX = 448.0
BLOCK = 32


def mx_quantize(x: torch.Tensor, row_scaled: bool = False):
    """x [rows, K] (K % 32 == 0), any float dtype -> (q uint8 [rows, K] e4m3 bits, e8m0 uint8 [rows, K/32], row_scales fp32 [rows] | None).

    row_scaled (weights): first y = x / row_scale, row_scale = max|row| / 448 in fp32 (1 for an all-zero row); then blocks of y."""
    xf = x.detach().float()
    rows, K = xf.shape
    assert K % BLOCK == 0
    rs = None
    if row_scaled:
        amax = xf.abs().amax(1)
        rs = torch.where(amax > 0, amax * torch.tensor(1.0 / 448.0), torch.ones_like(amax))
        xf = xf * (1.0 / rs)[:, None]                           # the kernel multiplies by the fp32 reciprocal
    blk = xf.view(rows, K // BLOCK, BLOCK)
    a = blk.abs().amax(2)
    t = (a * torch.tensor(1.0 / 448.0)).float()                  # fp32 product, as on the device
    m, e = torch.frexp(t)                                        # t = m * 2^e, m in [0.5, 1)
    E = torch.where(m == 0.5, e - 1, e)                          # ceil(log2 t): exact powers of two keep their exponent
    byte = torch.where(a > 0, (E + 127).clamp(1, 254), torch.full_like(E, 127)).to(torch.int32)
    inv = torch.ldexp(torch.ones_like(t), (127 - byte))          # 2^-(byte - 127)
    q = (blk * inv[:, :, None]).to(torch.float8_e4m3fn).view(torch.uint8).view(rows, K)
    return q, byte.to(torch.uint8), rs


def mx_dequantize(q: torch.Tensor, e8m0: torch.Tensor, row_scales=None) -> torch.Tensor:
    """float64 values the quantised operand stands for."""
    rows, K = q.shape
    v = q.view(torch.float8_e4m3fn).double().view(rows, K // BLOCK, BLOCK)
    v = v * torch.ldexp(torch.ones(e8m0.shape, dtype=torch.float64), e8m0.to(torch.int32) - 127)[:, :, None]
    v = v.view(rows, K)
    return v * row_scales.double()[:, None] if row_scales is not None else v


def mx_scale_layout(e8m0: torch.Tensor) -> torch.Tensor:
    """[rows, K/32] block-major scales -> the library's byte layout [rows, 4 * nkp] (include/titok_hip.h, ttv_quant_mx_fp8): block b at
    (b & 3) * nkp + (b >> 2), nkp = round_up(K / 128, 4); unused bytes 0."""
    rows, nb = e8m0.shape
    nk = nb // 4
    nkp = (nk + 3) // 4 * 4
    out = torch.zeros(rows, 4, nkp, dtype=torch.uint8)
    out[:, :, :nk] = e8m0.view(rows, nk, 4).transpose(1, 2)
    return out.view(rows, 4 * nkp)
