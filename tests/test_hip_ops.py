"""Per-kernel parity: every C-ABI op of libtitok_hip.so against the CPU oracle on the same seeded inputs.

All tests need the MI355X (`-m gpu`).  Tolerances are written next to each comparison:
  * integer outputs (FSQ indices, histogram) are bit-exact;
  * fp32 kernels: differences come from summation order / libm only;
  * bf16 kernels: inputs are bf16-rounded once for both sides; the kernel accumulates in fp32 and rounds its
    output to bf16 once, so the bound is a few bf16 ulps (2^-8 relative) of the output magnitude.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import titok_oracle as O
from titok_video_amd import _lib
from titok_video_amd.model.quantizer.fsq import FSQ
from titok_video_amd.plan import BatchPlan
from titok_video_amd.synthetic import seeded_titok_state

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
DT = {"bf16": torch.bfloat16, "f32": torch.float32}


def L():
    return _lib.lib()


def S():
    return _lib.stream_ptr(torch.device(DEV))


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def tol(dt):   # (relative Frobenius error, max-abs error as a fraction of max|ref|)
    return (3e-3, 2e-2) if dt == "bf16" else (2e-6, 2e-5)


def assert_close(out, ref, dt, scale=1.0):
    r, m = tol(dt)
    ref = ref.double().cpu()
    out = out.double().cpu()
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < r * scale, rel_err(out, ref)
    assert float((out - ref).abs().max()) <= m * scale * float(ref.abs().max() + 1e-30)


# ---------------------------------------------------------------------------------------------- FSQ
@pytest.mark.parametrize("tag", ["a", "b"])
def test_fsq_kat_bit_exact(tag):
    d = np.load(os.path.join(G, "fsq_kat.npz"))
    levels = d[f"levels_{tag}"].tolist()
    f = FSQ(levels)
    z = torch.from_numpy(d[f"z_{tag}"]).to(DEV)
    codes, dd = f(z)
    idx = dd["indices"].cpu().numpy()
    bounded = f.bounded(z).cpu()
    ref_idx, ref_b = d[f"indices_{tag}"], torch.from_numpy(d[f"bounded_{tag}"])
    # tanhf on the device and torch's CPU tanh may differ in the last ulp: bounded agrees to 2 ulp of half_l ...
    assert float((bounded - ref_b).abs().max()) < 2e-6
    # ... so an index may only differ where the reference value sits within that distance of a rounding boundary
    margin = O.fsq_margin(ref_b).numpy()
    safe = margin > 4e-6
    assert np.array_equal(idx[safe], ref_idx[safe])
    # the mismatches that remain (device tanhf vs the CPU's tanh in the last ulp, exactly at a rounding boundary) are listed and
    # pinned: at most two of the sweep's values, each within 4e-6 of a boundary
    bad = np.nonzero(idx != ref_idx)[0]
    for i in bad:
        print(f"fsq kat {tag}: z = {z[i].cpu().numpy()} index {idx[i]} != reference {ref_idx[i]}; bounded {bounded[i].numpy()} vs {ref_b[i].numpy()}; margin {margin[i]:.2e}")
    print(f"fsq kat {tag}: {len(bad)} of {len(idx)} indices differ from the reference")
    assert len(bad) <= 2 and all(margin[i] <= 4e-6 for i in bad)
    assert np.array_equal(codes.cpu().numpy()[safe], d[f"codes_{tag}"][safe])
    n = int(np.prod(levels))
    cb = f.indices_to_codes(torch.arange(n, dtype=torch.int32, device=DEV)).cpu().numpy()
    assert np.array_equal(cb, d[f"codebook_{tag}"])
    assert np.array_equal(cb, f.implicit_codebook.numpy())
    # round trip: quantising a codebook entry (scaled into the bound's linear range) returns its index
    codes2, dd2 = f(torch.atanh(torch.from_numpy(cb).clamp(-0.999, 0.999)).to(DEV) * 0 + torch.from_numpy(cb).to(DEV) * 20)
    assert idx.min() >= 0 and idx.max() < n


def test_fsq_bf16_io_and_empty():
    f = FSQ([8, 8, 8, 6, 5])
    g = torch.Generator().manual_seed(1)
    z = (torch.randn(1000, 5, generator=g) * 1.5).to(torch.bfloat16)
    codes, dd = f(z.to(DEV))
    rc, ri, rb = O.fsq_forward(z, [8, 8, 8, 6, 5])
    safe = (O.fsq_margin(rb) > 1e-5).numpy()
    assert np.array_equal(dd["indices"].cpu().numpy()[safe], ri.numpy()[safe])
    assert torch.equal(codes.cpu()[torch.from_numpy(safe)], rc[torch.from_numpy(safe)])
    c0, d0 = f(torch.empty(0, 5, device=DEV))
    assert c0.shape == (0, 5) and d0["indices"].shape == (0,)


# ---------------------------------------------------------------------------------------------- RMSNorm
@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("d", [256, 768, 1024])
def test_rmsnorm(dt, d):
    g = torch.Generator().manual_seed(d)
    rows = 37
    x = (torch.randn(rows, d, generator=g) * 3).to(DT[dt])
    w = 1 + 0.1 * torch.randn(d, generator=g)
    src = torch.randperm(rows, generator=g).to(torch.int32)
    dst = torch.randperm(rows + 5, generator=g)[:rows].to(torch.int32)
    out = torch.zeros(rows + 5, d, dtype=DT[dt], device=DEV)
    xd, wd, sd, dd = x.to(DEV), w.to(DEV), src.to(DEV), dst.to(DEV)
    code = _lib.dtype_code(DT[dt])
    _lib.check(L().ttv_rmsnorm(xd.data_ptr(), code, d, sd.data_ptr(), out.data_ptr(), code, d, dd.data_ptr(), wd.data_ptr(),
                               rows, d, 1e-5, S()), "rmsnorm")
    ref = torch.zeros(rows + 5, d)
    ref[dst.long()] = O.rmsnorm(x[src.long()], w).float()
    assert_close(out.float(), ref, dt)
    # fp32 in -> bf16 out (the KEEL post-norm path)
    if dt == "bf16":
        xf = torch.randn(rows, d, generator=g) * 10
        o2 = torch.empty(rows, d, dtype=torch.bfloat16, device=DEV)
        xfd = xf.to(DEV)
        _lib.check(L().ttv_rmsnorm(xfd.data_ptr(), _lib.TTV_F32, d, None, o2.data_ptr(), _lib.TTV_BF16, d, None, wd.data_ptr(),
                                   rows, d, 1e-5, S()), "rmsnorm")
        assert_close(o2.float(), O.rmsnorm(xf, w), "bf16")


@pytest.mark.parametrize("rows", [1, 7, 8, 33, 1000, 4099])
def test_rmsnorm_width_256_bf16_fast_path(rows):
    """k_rmsnorm256_bf16 (half a wave per row, eight rows of a wave in flight; taken for bf16 -> bf16 at width 256 without row maps): ragged
    row counts (the last wave's rows past the end are clamped loads, no stores), leading dimensions larger than the width, against the
    oracle and - through identity row maps, which keep the one-row-per-wave kernel - against the general kernel to one bf16 ulp."""
    g = torch.Generator().manual_seed(rows)
    d, ld_in, ld_out = 256, 264, 320
    x = torch.zeros(rows, ld_in).to(torch.bfloat16)
    x[:, :d] = (torch.randn(rows, d, generator=g) * 3).to(torch.bfloat16)
    w = 1 + 0.1 * torch.randn(d, generator=g)
    xd, wd = x.to(DEV), w.to(DEV)
    out = torch.full((rows + 2, ld_out), 7.0, dtype=torch.bfloat16, device=DEV)
    _lib.check(L().ttv_rmsnorm(xd.data_ptr(), _lib.TTV_BF16, ld_in, None, out.data_ptr(), _lib.TTV_BF16, ld_out, None, wd.data_ptr(), rows, d, 1e-5, S()),
               "rmsnorm")
    assert_close(out[:rows, :d].float(), O.rmsnorm(x[:, :d], w).float(), "bf16")
    assert bool((out[rows:] == 7.0).all()) and bool((out[:, d:] == 7.0).all())          # nothing written past the rows / the width
    ident = torch.arange(rows, dtype=torch.int32, device=DEV)
    gen = torch.empty(rows, d, dtype=torch.bfloat16, device=DEV)
    _lib.check(L().ttv_rmsnorm(xd.data_ptr(), _lib.TTV_BF16, ld_in, ident.data_ptr(), gen.data_ptr(), _lib.TTV_BF16, d, None, wd.data_ptr(), rows, d, 1e-5,
                               S()), "rmsnorm")
    diff = (out[:rows, :d].float() - gen.float()).abs()
    assert float((diff / gen.float().abs().clamp_min(1e-3)).max()) <= 2.0 ** -7


# ---------------------------------------------------------------------------------------------- RoPE
@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_rope_apply_matches_reference_fixture(dt):
    d = np.load(os.path.join(G, "rope_kat.npz"))
    grids, counts = d["grids_1"].tolist(), d["counts_1"].tolist()
    plan = BatchPlan([(g[0] * 4, g[1] * 8, g[2] * 8) for g in grids], counts, (4, 8, 8), DEV)
    # host table == reference freqs_cis (fp64 -> fp32)
    cs = plan.rope_cs.cpu().numpy()
    assert np.array_equal(cs[:, :30], d["cos_1"].astype(np.float32))
    assert np.array_equal(cs[:, 32:62], d["sin_1"].astype(np.float32))
    assert np.all(cs[:, 30:32] == 1) and np.all(cs[:, 62:] == 0)
    q = torch.from_numpy(d["rot_q"]).to(DT[dt])
    qd = q.to(DEV).contiguous()
    _lib.check(L().ttv_rope_apply(qd.data_ptr(), _lib.dtype_code(DT[dt]), 256, q.shape[0], 4, plan.rope_cs.data_ptr(), S()), "rope")
    if dt == "f32":
        np.testing.assert_allclose(qd.cpu().numpy(), d["rot_out"], rtol=0, atol=2e-6)
        assert np.array_equal(qd.cpu().numpy()[..., 60:], d["rot_q"][..., 60:])
    else:
        cos, sin = O.rope_table(grids, counts)
        assert_close(qd.float(), O.apply_rotary(q.float(), cos, sin), "bf16")


# ---------------------------------------------------------------------------------------------- linear layers
def _lin_inputs(M, N, K, dt, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).to(DT[dt])
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DT[dt])
    return x, w, g


@pytest.mark.parametrize("tile", [128, 160])
@pytest.mark.parametrize("shape", [(300, 256, 768), (161, 768, 704), (1, 256, 704), (513, 256, 1376), (4000, 256, 1408)])
def test_linear_general_k_tile_heights(shape, tile):
    """General-K bf16 GEMM with the 128- and the 160-token tile forced (the host picks by grid balance: 36 864 x 256 outputs are
    576 tiles of 128 rows = two rounds of 512 resident blocks, 462 tiles of 160 rows = one)."""
    M, N, K = shape
    x, w, g = _lin_inputs(M, N, K, "bf16", M + N)
    y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    xd, wd = x.to(DEV), w.to(DEV)
    L().ttv_debug_set(128 if tile == 160 else 256)
    try:
        _lib.check(L().ttv_linear(xd.data_ptr(), K, wd.data_ptr(), K, None, None, y.data_ptr(), N, M, N, K, _lib.dtype_code(torch.bfloat16), S()), "linear")
        torch.cuda.synchronize()
    finally:
        L().ttv_debug_set(0)
    assert_close(y.float(), x.double() @ w.double().T, "bf16")


@pytest.mark.parametrize("case", ["store", "geglu", "qkv_rope", "resid_bf16", "resid_f32"])
@pytest.mark.parametrize("M", [1, 255, 300, 1025])
def test_linear_256x256_tiles(case, M):
    """k_gemm_bf16_t256 (256-feature x 256-token tiles, 8 waves; forced with ttv_debug_set bit 512) under every epilogue it serves,
    against fp64 and against the 128 x 128 kernel (bit 1024 forbids the large tile)."""
    g = torch.Generator().manual_seed(M + len(case))
    bf = torch.bfloat16
    code = _lib.dtype_code(bf)
    outs = []
    for bit in (512, 1024):
        L().ttv_debug_set(bit)
        try:
            if case == "store":
                N, K = 512, 768
                x, w, _ = _lin_inputs(M, N, K, "bf16", 3)
                y = torch.full((M, N), float("nan"), dtype=bf, device=DEV)
                xd, wd = x.to(DEV), w.to(DEV)
                _lib.check(L().ttv_linear(xd.data_ptr(), K, wd.data_ptr(), K, None, None, y.data_ptr(), N, M, N, K, code, S()), "linear")
                ref = x.double() @ w.double().T
            elif case == "geglu":
                I, K = 384, 704
                x, w, _ = _lin_inputs(M, 2 * I, K, "bf16", 4)
                y = torch.full((M, I), float("nan"), dtype=bf, device=DEV)
                xd, wd = x.to(DEV), w.to(DEV)
                _lib.check(L().ttv_linear_geglu(xd.data_ptr(), K, wd.data_ptr(), K, y.data_ptr(), I, M, I, K, code, S()), "geglu")
                h = x.double() @ w.double().T
                a, gate = h.chunk(2, -1)
                ref = torch.nn.functional.gelu(gate) * a
            elif case == "qkv_rope":
                plan = BatchPlan([(8, 32, 48), (4, 16, 24)], [3, 5], (4, 8, 8), DEV)
                Mq, d, gq = plan.total_rows, 768, 256
                x, w, _ = _lin_inputs(Mq, 2 * d + 2 * gq, d, "bf16", 5)
                y = torch.full((Mq, 2 * d + 2 * gq), float("nan"), dtype=bf, device=DEV)
                xd, wd = x.to(DEV), w.to(DEV)
                _lib.check(L().ttv_linear_qkv_rope(xd.data_ptr(), d, wd.data_ptr(), d, y.data_ptr(), 2 * d + 2 * gq, Mq, d, gq,
                                                   plan.rope_cs.data_ptr(), code, S()), "qkv")
                r0 = (x.double() @ w.double().T).float()
                q, gate, k, v = r0.split([d, d, gq, gq], dim=-1)
                cos, sin = O.rope_table(plan.grids, plan.token_counts)
                q = O.apply_rotary(q.unflatten(-1, (12, 64)), cos, sin).flatten(-2)
                k = O.apply_rotary(k.unflatten(-1, (4, 64)), cos, sin).flatten(-2)
                ref = torch.cat([q, gate, k, v], -1)
            else:
                f32out = case == "resid_f32"
                N, K = 768, 2048
                x, w, _ = _lin_inputs(M, N, K, "bf16", 6)
                r = torch.randn(M, N, generator=torch.Generator().manual_seed(M)).to(bf)
                y = torch.full((M, N), float("nan"), dtype=torch.float32 if f32out else bf, device=DEV)
                xd, wd, rd = x.to(DEV), w.to(DEV), r.to(DEV)
                _lib.check(L().ttv_linear_residual(xd.data_ptr(), K, wd.data_ptr(), K, rd.data_ptr(), N, 8.0, y.data_ptr(), N, int(f32out), M, N, K, code, S()), "resid")
                ref = 8.0 * r.double() + x.double() @ w.double().T
            torch.cuda.synchronize()
        finally:
            L().ttv_debug_set(0)
        assert_close(y.float(), ref, "bf16")
        outs.append(y.clone())
    assert torch.equal(outs[0], outs[1])     # same products in the same k order, same epilogue: bit-equal to the 128 x 128 kernel


@pytest.mark.parametrize("K", [128, 192, 256])
@pytest.mark.parametrize("M", [255, 513])
def test_linear_256x256_tiles_short_k(K, M):
    """nk = 2, 3, 4 k-tiles (the kernel needs K >= 128): with two tiles the prologue stages everything and no piece is issued inside
    the loop; with three, tile 2 is staged by the loop's phase 4 / phase 2 path exactly once.  Bit-equal to the 128 x 128 kernel."""
    N = 512
    x, w, _ = _lin_inputs(M, N, K, "bf16", 17 + K)
    xd, wd = x.to(DEV), w.to(DEV)
    code = _lib.dtype_code(torch.bfloat16)
    outs = []
    for bit in (512, 1024):
        y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        L().ttv_debug_set(bit)
        try:
            _lib.check(L().ttv_linear(xd.data_ptr(), K, wd.data_ptr(), K, None, None, y.data_ptr(), N, M, N, K, code, S()), "linear")
            torch.cuda.synchronize()
        finally:
            L().ttv_debug_set(0)
        assert_close(y.float(), x.double() @ w.double().T, "bf16")
        outs.append(y)
    assert torch.equal(outs[0], outs[1])


def test_linear_256x256_tiles_repeatable():
    """Race screen of k_gemm_bf16_t256's LDS-DMA / barrier schedule (the long version is tools/gemm_t256_bench.py): the same launch
    many times over a grid that fills the part several times, every result bit-equal to the 128 x 128 kernel's."""
    M, N, K = 9216, 2048, 768          # 36 x 8 = 288 tiles of 256 x 256, 12 k-tiles
    x, w, _ = _lin_inputs(M, N, K, "bf16", 11)
    xd, wd = x.to(DEV), w.to(DEV)
    code = _lib.dtype_code(torch.bfloat16)
    y0 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    y1 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    try:
        L().ttv_debug_set(1024)
        _lib.check(L().ttv_linear(xd.data_ptr(), K, wd.data_ptr(), K, None, None, y0.data_ptr(), N, M, N, K, code, S()), "linear")
        L().ttv_debug_set(512)
        for _ in range(40):
            y1.fill_(float("nan"))
            _lib.check(L().ttv_linear(xd.data_ptr(), K, wd.data_ptr(), K, None, None, y1.data_ptr(), N, M, N, K, code, S()), "linear")
            assert torch.equal(y0, y1)
    finally:
        L().ttv_debug_set(0)


@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("shape", [(300, 256, 768), (129, 768, 256), (64, 8, 256), (1, 256, 704), (513, 256, 1376)])
def test_linear_bias_scalar(dt, shape):
    M, N, K = shape
    x, w, g = _lin_inputs(M, N, K, dt, M + N)
    b = (torch.randn(N, generator=g) * 0.1).to(DT[dt])
    sc = torch.tensor([0.37])
    y = torch.empty(M, N, dtype=DT[dt], device=DEV)
    xd, wd, bd, sd = x.to(DEV), w.to(DEV), b.to(DEV), sc.to(DEV)
    _lib.check(L().ttv_linear(xd.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), sd.data_ptr(), y.data_ptr(), N, M, N, K,
                              _lib.dtype_code(DT[dt]), S()), "linear")
    ref = x.double() @ w.double().T + b.double() + float(sc.to(DT[dt]))
    assert_close(y.float(), ref, dt)


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_linear_qkv_rope(dt):
    plan = BatchPlan([(8, 32, 48), (4, 16, 24)], [3, 5], (4, 8, 8), DEV)
    M, d, gq = plan.total_rows, 256, 128
    x, w, g = _lin_inputs(M, 2 * d + 2 * gq, d, dt, 5)
    y = torch.empty(M, 2 * d + 2 * gq, dtype=DT[dt], device=DEV)
    xd, wd = x.to(DEV), w.to(DEV)
    _lib.check(L().ttv_linear_qkv_rope(xd.data_ptr(), d, wd.data_ptr(), d, y.data_ptr(), 2 * d + 2 * gq, M, d, gq,
                                       plan.rope_cs.data_ptr(), _lib.dtype_code(DT[dt]), S()), "qkv")
    ref = (x.double() @ w.double().T).float()
    q, gate, k, v = ref.split([d, d, gq, gq], dim=-1)
    cos, sin = O.rope_table(plan.grids, plan.token_counts)
    q = O.apply_rotary(q.unflatten(-1, (4, 64)), cos, sin).flatten(-2)
    k = O.apply_rotary(k.unflatten(-1, (2, 64)), cos, sin).flatten(-2)
    assert_close(y.float(), torch.cat([q, gate, k, v], -1), dt)


@pytest.mark.parametrize("shape", ["ragged", "one_tile", "many_tiles"])
def test_to_qkv_width256_kernels_give_the_same_bits(shape):
    """The three kernels behind `ttv_linear_qkv_rope` at K = 256 bf16 - k_qkv256 (default: wave-pipelined, 32x32x16 MFMAs), k_qkv256ws
    (weights stationary in LDS, ttv_debug_set bit 17) and k_gemm_k256<EPI_QKV_ROPE> (bit 15) - multiply the same operands in the same
    k order and rotate with the same fp32 arithmetic: without the folded pre-norm their outputs are bit-identical, whatever the
    decomposition (ragged last tile, fewer rows than one tile, several tiles per block)."""
    shapes, counts = {"ragged": ([(8, 32, 48), (4, 16, 24), (4, 24, 40)], [3, 5, 7]), "one_tile": ([(4, 16, 16)], [2]),
                      "many_tiles": ([(16, 64, 64)] * 5, [32] * 5)}[shape]
    plan = BatchPlan(shapes, counts, (4, 8, 8), DEV)
    M, d, gq = plan.total_rows, 256, 128
    x, w, _ = _lin_inputs(M, 2 * d + 2 * gq, d, "bf16", 11)
    xd, wd = x.to(DEV), w.to(DEV)
    outs = []
    try:
        for bits in (0, 1 << 15, 1 << 17):
            y = torch.full((M, 2 * d + 2 * gq), float("nan"), dtype=torch.bfloat16, device=DEV)
            L().ttv_debug_set(bits)
            _lib.check(L().ttv_linear_qkv_rope(xd.data_ptr(), d, wd.data_ptr(), d, y.data_ptr(), 2 * d + 2 * gq, M, d, gq,
                                               plan.rope_cs.data_ptr(), _lib.dtype_code(torch.bfloat16), S()), "qkv")
            torch.cuda.synchronize()
            outs.append(y)
    finally:
        L().ttv_debug_set(0)
    assert not torch.isnan(outs[0].float()).any()
    assert torch.equal(outs[0], outs[1]), "k_qkv256 differs from k_gemm_k256"
    assert torch.equal(outs[2], outs[1]), "k_qkv256ws differs from k_gemm_k256"


@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("I", [704, 1376, 96])
def test_linear_geglu(dt, I):
    M, K = 257, 256
    x, w, g = _lin_inputs(M, 2 * I, K, dt, I)
    y = torch.empty(M, I, dtype=DT[dt], device=DEV)
    xd, wd = x.to(DEV), w.to(DEV)
    _lib.check(L().ttv_linear_geglu(xd.data_ptr(), K, wd.data_ptr(), K, y.data_ptr(), I, M, I, K, _lib.dtype_code(DT[dt]), S()), "geglu")
    h = x.double() @ w.double().T
    a, gate = h.chunk(2, -1)
    assert_close(y.float(), torch.nn.functional.gelu(gate) * a, dt)


@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("f32out", [0, 1])
def test_linear_residual(dt, f32out):
    M, N, K = 200, 256, 704
    x, w, g = _lin_inputs(M, N, K, dt, 9)
    r = torch.randn(M, N, generator=g).to(DT[dt])
    y = torch.empty(M, N, dtype=torch.float32 if f32out else DT[dt], device=DEV)
    xd, wd, rd = x.to(DEV), w.to(DEV), r.to(DEV)
    _lib.check(L().ttv_linear_residual(xd.data_ptr(), K, wd.data_ptr(), K, rd.data_ptr(), N, 8.0, y.data_ptr(), N, f32out, M, N, K,
                                       _lib.dtype_code(DT[dt]), S()), "resid")
    ref = 8.0 * r.double() + x.double() @ w.double().T
    assert_close(y.float(), ref, "f32" if (f32out and dt == "f32") else dt)
    if not f32out:   # in place on the residual (layer-0 path)
        _lib.check(L().ttv_linear_residual(xd.data_ptr(), K, wd.data_ptr(), K, rd.data_ptr(), N, 1.0, rd.data_ptr(), N, 0, M, N, K,
                                           _lib.dtype_code(DT[dt]), S()), "resid")
        assert_close(rd.float(), r.double() + x.double() @ w.double().T, dt)


@pytest.mark.parametrize("M", [1, 63, 64, 200, 4097])
@pytest.mark.parametrize("K", [256, 704, 1376])
def test_linear_residual_norm_fused(M, K):
    """KEEL step in one kernel (bf16, N = 256): y = RMSNorm(alpha*resid + x@w^T)*gain, in place on resid.
    K = 256: register-resident-token kernel (out_proj); other K: full-row tile kernel (w3)."""
    N = 256
    x, w, g = _lin_inputs(M, N, K, "bf16", 17 + M)
    r = torch.randn(M, N, generator=g).to(torch.bfloat16)
    gain = 1 + 0.1 * torch.randn(N, generator=g)
    xd, wd, rd, gd = x.to(DEV), w.to(DEV), r.to(DEV), gain.to(DEV)
    _lib.check(L().ttv_linear_residual_norm(xd.data_ptr(), K, wd.data_ptr(), K, rd.data_ptr(), N, 8.0, gd.data_ptr(), 1e-5,
                                            rd.data_ptr(), N, M, N, K, _lib.TTV_BF16, S()), "resid_norm")
    y = 8.0 * r.double() + x.double() @ w.double().T
    ref = y * torch.rsqrt(y.pow(2).mean(-1, keepdim=True) + 1e-5) * gain.double()
    assert_close(rd.float(), ref, "bf16")
    rc = L().ttv_linear_residual_norm(xd.data_ptr(), K, wd.data_ptr(), K, rd.data_ptr(), N, 8.0, gd.data_ptr(), 1e-5,
                                      rd.data_ptr(), N, M, N, K, _lib.TTV_F32, S())
    assert rc == 3      # TTV_ERR_UNSUPPORTED: callers fall back to linear_residual + rmsnorm


@pytest.mark.parametrize("M", [1, 100, 192, 200, 1000, 20000, 50000])   # 64-, 128- and 192-token tile variants
@pytest.mark.parametrize("keel", [True, False])
@pytest.mark.parametrize("I", [704, 96, 32])
@pytest.mark.parametrize("deal9", [False, True])                        # 144-token blocks dealt over the eight waves (ttv_debug bit 9)
def test_mlp_fused(M, keel, I, deal9):
    """Whole GEGLU sub-layer + residual/KEEL in one kernel vs the op-by-op definition (transformer.py:47-56,130,144-145)."""
    if deal9 and I != 704 and M not in (100, 1000):
        pytest.skip("the 9-tile deal is exercised at I = 704 and two ragged sizes of the other widths")
    d = 256
    g = torch.Generator().manual_seed(M + I)
    x = (torch.randn(M, d, generator=g) * 1.3).to(torch.bfloat16)
    w12 = (torch.randn(2 * I, d, generator=g) * d ** -0.5).to(torch.bfloat16)
    w3 = (torch.randn(d, I, generator=g) * I ** -0.5).to(torch.bfloat16)
    ng = 1 + 0.1 * torch.randn(d, generator=g)
    pg = 1 + 0.1 * torch.randn(d, generator=g)
    w12f = (w12.float() * ng[None, :]).to(torch.bfloat16)
    xd, w12d, w3d, pgd = x.to(DEV), w12f.to(DEV), w3.to(DEV), pg.to(DEV)
    pack = torch.empty(L().ttv_mlp_pack_bytes(I, 0), dtype=torch.uint8, device=DEV)
    assert pack.numel() == (I // 32) * 48 * 1024 + 128 * 1024
    _lib.check(L().ttv_mlp_pack(w12d.data_ptr(), w3d.data_ptr(), None, None, 0, I, d, _lib.TTV_BF16, pack.data_ptr(), S()), "mlp_pack")
    alpha = 8.0 if keel else 1.0
    L().ttv_debug_set(512 if deal9 else 0)
    try:
        _lib.check(L().ttv_mlp_fused(xd.data_ptr(), d, pack.data_ptr(), I, xd.data_ptr(), d,
                                     pgd.data_ptr() if keel else None, alpha, 1e-5, M, d, _lib.TTV_BF16, S()), "mlp_fused")
        torch.cuda.synchronize()
    finally:
        L().ttv_debug_set(0)
    xf = x.double()
    xn = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)
    hh = xn @ w12f.double().T
    a, gate = hh.chunk(2, -1)
    h = (torch.nn.functional.gelu(gate) * a).to(torch.bfloat16).double()      # the kernel rounds h to bf16 (MFMA operand)
    y = alpha * xf + h @ w3.double().T
    ref = y * torch.rsqrt(y.pow(2).mean(-1, keepdim=True) + 1e-5) * pg.double() if keel else y
    assert_close(xd.float(), ref, "bf16", scale=1.5)
    assert L().ttv_mlp_fused(xd.data_ptr(), d, pack.data_ptr(), I, xd.data_ptr(), d, None, 1.0, 1e-5, M, 512,
                             _lib.TTV_BF16, S()) == 3


@pytest.mark.parametrize("M", [1, 100, 192, 1000, 20000, 36864, 40000])   # 36864: the benchmark batch picks the 9-tile deal by itself
@pytest.mark.parametrize("keel", [True, False])
@pytest.mark.parametrize("back", [False, True, "deal9"])                  # "deal9": 144-token blocks forced (ttv_debug bit 9), no fused QKV
def test_layer_tail_fused(M, keel, back):
    """out_proj + residual/KEEL + GEGLU sub-layer + residual/KEEL (+ the next layer's pre_ln + to_qkv + rotary) in one kernel
    vs the op-by-op definition (transformer.py:104,129-130 / 141-145, 47-56, 86-98)."""
    import ctypes as C
    deal9 = back == "deal9"
    back = back is True
    if M == 36864 and (back or deal9):
        pytest.skip("the benchmark size runs once per gain setting")
    d, I, gq = 256, 704, 128
    nq = 2 * d + 2 * gq
    g = torch.Generator().manual_seed(M + 7)
    x = (torch.randn(M, d, generator=g) * 1.3).to(torch.bfloat16)
    ao = (torch.randn(M, d, generator=g)).to(torch.bfloat16)
    wo = (torch.randn(d, d, generator=g) * d ** -0.5).to(torch.bfloat16)
    w12 = (torch.randn(2 * I, d, generator=g) * d ** -0.5).to(torch.bfloat16)
    w3 = (torch.randn(d, I, generator=g) * I ** -0.5).to(torch.bfloat16)
    wq = (torch.randn(nq, d, generator=g) * d ** -0.5).to(torch.bfloat16)
    ng, ag, pg, qg = (1 + 0.1 * torch.randn(d, generator=g) for _ in range(4))
    ang = torch.rand(M, 32, generator=g) * 6.28
    cs = torch.cat([ang.cos(), ang.sin()], 1).float()
    w12f = (w12.float() * ng[None, :]).to(torch.bfloat16)
    wqf = (wq.float() * qg[None, :]).to(torch.bfloat16)
    xd, aod, w12d, w3d, wod, wqd, agd, pgd, csd = (t.to(DEV) for t in (x, ao, w12f, w3, wo, wqf, ag, pg, cs))
    rows = nq if back else 0
    pack = torch.empty(L().ttv_mlp_pack_bytes(I, rows), dtype=torch.uint8, device=DEV)
    _lib.check(L().ttv_mlp_pack(w12d.data_ptr(), w3d.data_ptr(), wod.data_ptr(), wqd.data_ptr() if back else None, rows, I, d,
                                _lib.TTV_BF16, pack.data_ptr(), S()), "mlp_pack")
    qkv = torch.zeros(M, nq, dtype=torch.bfloat16, device=DEV)
    nx = _lib.NextQkv(qkv=qkv.data_ptr(), ld=nq, rope_cs=csd.data_ptr(), rows=nq, rope_q_end=d, rope_k_begin=2 * d, rope_k_end=2 * d + gq)
    alpha = 8.0 if keel else 1.0
    L().ttv_debug_set(512 if deal9 else 0)
    try:
        _lib.check(L().ttv_layer_tail_fused(aod.data_ptr(), d, agd.data_ptr() if keel else None, alpha, xd.data_ptr(), d, pack.data_ptr(), I,
                                            xd.data_ptr(), d, pgd.data_ptr() if keel else None, alpha, 1e-5, M, d, _lib.TTV_BF16,
                                            C.byref(nx) if back else None, S()), "layer_tail_fused")
        torch.cuda.synchronize()
    finally:
        L().ttv_debug_set(0)
    y1 = alpha * x.double() + ao.double() @ wo.double().T
    x1 = y1 * torch.rsqrt(y1.pow(2).mean(-1, keepdim=True) + 1e-5) * ag.double() if keel else y1
    x1 = x1.to(torch.bfloat16).double()                                      # the kernel rounds x1 to bf16 (residual stream dtype)
    xn = x1 * torch.rsqrt(x1.pow(2).mean(-1, keepdim=True) + 1e-5)
    a, gate = (xn @ w12f.double().T).chunk(2, -1)
    h = (torch.nn.functional.gelu(gate) * a).to(torch.bfloat16).double()
    y = alpha * x1 + h @ w3.double().T
    ref = y * torch.rsqrt(y.pow(2).mean(-1, keepdim=True) + 1e-5) * pg.double() if keel else y
    assert_close(xd.float(), ref, "bf16", scale=2.0)
    if back:
        x2 = xd.double().cpu()                                               # the projection reads the stored (bf16) rows
        q = (x2 * torch.rsqrt(x2.pow(2).mean(-1, keepdim=True) + 1e-5)) @ wqf.double().T

        def rot(t):                                                          # interleaved pairs, per 64-wide head
            th = t.reshape(M, -1, 32, 2)
            c, sn = cs[:, None, :32].double(), cs[:, None, 32:].double()
            return torch.stack([th[..., 0] * c - th[..., 1] * sn, th[..., 0] * sn + th[..., 1] * c], -1).reshape(M, -1)
        qref = torch.cat([rot(q[:, :d]), q[:, d:2 * d], rot(q[:, 2 * d:2 * d + gq]), q[:, 2 * d + gq:]], 1)
        assert_close(qkv.float(), qref, "bf16", scale=2.0)


# ---------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("case", [([(4, 16, 16)], [1]), ([(8, 32, 48), (4, 8, 24), (16, 64, 64)], [5, 3, 128]),
                                  ([(16, 128, 128)], [128]), ([(4, 8, 8), (4, 8, 8)], [0, 63])])
@pytest.mark.parametrize("heads", [(4, 2), (12, 4)])
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("paired", [0, 2])
@pytest.mark.parametrize("qscaled", [0, 4, 4 | 8, 4 | 8 | 16])
def test_attention_varlen_gqa_gate(dt, case, heads, split, paired, qscaled):
    shapes, counts = case
    # TTV_ATTN_QSCALED | TTV_ATTN_ALLFULL: k_attn_swp (ttv_attn_swp.hip, the default for such tables); with TTV_ATTN_PIPE: k_attn_pipe
    if qscaled & 8 and (split or paired):
        pytest.skip("the pipelined kernels take full items, unpaired")
    plan = BatchPlan(shapes, counts, (4, 8, 8), DEV)
    hq, hkv = heads
    if paired and ((hq // hkv) % 2 or dt != "bf16"):
        pytest.skip("paired tables need an even number of q-heads per kv-head and the bf16 kernel")
    if qscaled and dt != "bf16":
        pytest.skip("pre-scaled q is a bf16 kernel option")
    d, gq = hq * 64, hkv * 64
    ld = 2 * d + 2 * gq
    g = torch.Generator().manual_seed(len(shapes) + hq)
    qkvg = torch.randn(plan.total_rows, ld, generator=g)
    qkvg[:, :d] *= 2.0            # sharper softmax
    q_f32 = qkvg[:, :d].clone()
    qkvg = qkvg.to(DT[dt])
    out = torch.empty(plan.total_rows, d, dtype=DT[dt], device=DEV)
    xd = qkvg.to(DEV)
    if qscaled:      # TTV_ATTN_QSCALED: q carries head_dim^-0.5 * log2(e), applied before the one rounding to bf16
        xd[:, :d] = (q_f32 * (0.125 * 1.4426950408889634)).to(DT[dt]).to(DEV)
    for gate in (1, 0):
        tab = plan.attention_table(hq, hkv, split)     # 128-query items / 64-query half items (key range split in-block)
        out.fill_(float("nan"))
        _lib.check(L().ttv_attention(xd.data_ptr(), ld, out.data_ptr(), d, plan.cu_dev.data_ptr(), tab.data_ptr(),
                                     tab.shape[0], hq, hkv, 64, gate | paired | qscaled, _lib.dtype_code(DT[dt]), S()), "attention")
        f = qkvg.float()
        q, gt, k, v = f.split([d, d, gq, gq], dim=-1)
        ref = O.attention_varlen(q.unflatten(-1, (hq, 64)), k.unflatten(-1, (hkv, 64)), v.unflatten(-1, (hkv, 64)),
                                 plan.cu_seqlens).flatten(-2)
        if gate:
            ref = ref * torch.sigmoid(gt)
        assert_close(out.float(), ref, dt, scale=2.0 if dt == "bf16" else 5.0)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("qscaled", [0, 4, 4 | 8, 4 | 8 | 16])
@pytest.mark.parametrize("spike", [6.0, 30.0, 60.0])
@pytest.mark.parametrize("spike_key", [200, 204, 70])
def test_attention_online_softmax_rescale_branch(split, qscaled, spike, spike_key):
    """Force the running max to jump at a late key tile (spike one key against every query); with pre-scaled q the maximum lives
    inside the MFMA accumulator and the jump shifts the tile's scores, the running sums and the start vector.  spike 30: the
    jump is ~140 log2 units for the aligned query (exp2 against the old reference overflows to inf: the pipelined kernel's rare
    branch must take the raw scores again) and tens of units for the others."""
    plan = BatchPlan([(8, 32, 32)], [7], (4, 8, 8), DEV)   # S = 39 ... use a longer one
    plan = BatchPlan([(16, 64, 64)], [9], (4, 8, 8), DEV)  # S = 265 -> 5 key tiles
    if qscaled & 8 and split:
        pytest.skip("the pipelined kernel takes full items")
    hq, hkv, d, gq = 4, 2, 256, 128
    ld = 2 * d + 2 * gq
    g = torch.Generator().manual_seed(3)
    x = torch.randn(plan.total_rows, ld, generator=g) * 0.5
    q = x[:, :d].view(-1, 4, 64)
    # the spiked key dominates: 200 = 4th tile, held by the LOWER lane half of a query's lane pair; 204 by the UPPER half (the
    # row maximum must cross the halves: round 5 found it did not); 70 = the first tile behind the reference
    x[spike_key, 2 * d: 2 * d + gq] = spike * torch.sign(q[5, 0]).repeat(2)
    q_f32 = x[:, :d].clone()
    x = x.to(torch.bfloat16)
    out = torch.empty(plan.total_rows, d, dtype=torch.bfloat16, device=DEV)
    xd = x.to(DEV)
    if qscaled:
        xd[:, :d] = (q_f32 * (0.125 * 1.4426950408889634)).to(torch.bfloat16).to(DEV)
    tab = plan.attention_table(hq, hkv, split)
    _lib.check(L().ttv_attention(xd.data_ptr(), ld, out.data_ptr(), d, plan.cu_dev.data_ptr(), tab.data_ptr(),
                                 tab.shape[0], hq, hkv, 64, qscaled, _lib.TTV_BF16, S()), "attention")
    f = x.float()
    qq, gt, k, v = f.split([d, d, gq, gq], dim=-1)
    ref = O.attention_varlen(qq.unflatten(-1, (hq, 64)), k.unflatten(-1, (hkv, 64)), v.unflatten(-1, (hkv, 64)), plan.cu_seqlens).flatten(-2)
    assert_close(out.float(), ref, "bf16", scale=2.0)


@pytest.mark.parametrize("spike", [6.0, 30.0, 60.0])
@pytest.mark.parametrize("spike_keys", [(200,), (70,), (264,), (70, 130), (3,)])
def test_attention_swp_reference_shift_branch(spike, spike_keys):
    """k_attn_swp (ttv_attn_swp.hip) keeps the maximum of a row's first 64 scores as its softmax reference and checks only a tile's row
    sums; a tile whose sums leave [0, 2^30] sends the wave through the rare path (true maximum of the tile's raw scores - still in
    registers -, shift of O, l, both score sets in flight and the -m start vector, the tile's P again).  Forced here by spiking keys
    against one query direction: in a middle tile (200), in the first tile behind the reference (70: the check of iteration 1), in the
    masked last tile (264 of 265: the shift meets -inf scores), in two consecutive tiles (70 and 130, the second higher: the shifted
    next score set is shifted again), and inside the first tile (3: no shift at all, the reference already holds it).  spike 30 / 60:
    exp2 against the old reference overflows to inf for the aligned query (row sum inf or NaN), tens of log2 units for the others;
    6: below 2^30 for most rows - the branch is wave-uniform, rows that did not need it shift by 0."""
    plan = BatchPlan([(16, 64, 64)], [9], (4, 8, 8), DEV)  # S = 265 -> 5 key tiles, the last with 9 keys
    hq, hkv, d, gq = 4, 2, 256, 128
    ld = 2 * d + 2 * gq
    g = torch.Generator().manual_seed(3)
    x = torch.randn(plan.total_rows, ld, generator=g) * 0.5
    q = x[:, :d].view(-1, 4, 64)
    for n, key in enumerate(spike_keys):
        x[key, 2 * d: 2 * d + gq] = (spike + 8.0 * n) * torch.sign(q[5, 0]).repeat(2)
    x[140, :64] = q[5, 0]                 # the same direction for a query of another wave and block
    q_f32 = x[:, :d].clone()
    x = x.to(torch.bfloat16)
    out = torch.full((plan.total_rows, d), float("nan"), dtype=torch.bfloat16, device=DEV)
    xd = x.to(DEV)
    xd[:, :d] = (q_f32 * (0.125 * 1.4426950408889634)).to(torch.bfloat16).to(DEV)
    tab = plan.attention_table(hq, hkv, False)
    _lib.check(L().ttv_attention(xd.data_ptr(), ld, out.data_ptr(), d, plan.cu_dev.data_ptr(), tab.data_ptr(), tab.shape[0], hq, hkv, 64, 4 | 8,
                                 _lib.TTV_BF16, S()), "attention")
    f = x.float()
    qq, gt, k, v = f.split([d, d, gq, gq], dim=-1)
    ref = O.attention_varlen(qq.unflatten(-1, (hq, 64)), k.unflatten(-1, (hkv, 64)), v.unflatten(-1, (hkv, 64)), plan.cu_seqlens).flatten(-2)
    assert_close(out.float(), ref, "bf16", scale=2.0)


@pytest.mark.parametrize("level", [-40.0, -100.0, 90.0])
def test_attention_swp_rows_whose_scores_all_sit_far_from_zero(level):
    """k_attn_swp exponentiates the scores as they stand (no softmax reference in its loop) and reads off the row sums whether that was in
    range (2^-60 < l < 2^60); otherwise the block runs again through the exact loop.  Here EVERY score of some query rows is shifted by
    `level` log2-units (the keys share a direction u, those queries are level * u / |u|^2 plus noise): -40 stays in range (p ~ 2^-40,
    no fallback: the quotient must still be right), -100 underflows the row sums (l < 2^-60: fallback), +90 overflows them."""
    plan = BatchPlan([(16, 64, 64)], [9], (4, 8, 8), DEV)  # S = 265
    hq, hkv, d, gq = 4, 2, 256, 128
    ld = 2 * d + 2 * gq
    g = torch.Generator().manual_seed(11)
    x = torch.randn(plan.total_rows, ld, generator=g) * 0.3
    u = torch.randn(64, generator=g)
    u = u / u.norm() * 4.0                                        # |u| = 4
    x[:, 2 * d: 2 * d + gq] += u.repeat(2)                        # every key of both kv-heads carries u
    c_exp = 0.125 * 1.4426950408889634
    for row in (5, 77, 140, 264):                                  # queries of several waves / blocks, all four q-heads
        x[row, :d] += (level / (c_exp * 16.0)) * u.repeat(4)      # q . u * c_exp = level
    q_f32 = x[:, :d].clone()
    x = x.to(torch.bfloat16)
    out = torch.full((plan.total_rows, d), float("nan"), dtype=torch.bfloat16, device=DEV)
    xd = x.to(DEV)
    xd[:, :d] = (q_f32 * c_exp).to(torch.bfloat16).to(DEV)
    tab = plan.attention_table(hq, hkv, False)
    _lib.check(L().ttv_attention(xd.data_ptr(), ld, out.data_ptr(), d, plan.cu_dev.data_ptr(), tab.data_ptr(), tab.shape[0], hq, hkv, 64, 4 | 8,
                                 _lib.TTV_BF16, S()), "attention")
    # reference: float64 softmax attention on the operands the kernel sees - the PRE-SCALED q as rounded to bf16 (queries of magnitude
    # ~17 round differently before and after the factor, which alone moves these peaked rows by ~1 %: tests/probes/swp_level_probe.py)
    qs = xd[:, :d].double().cpu().view(-1, hq, 64)
    kk, vv = x[:, 2 * d:2 * d + gq].double().view(-1, hkv, 64), x[:, 2 * d + gq:].double().view(-1, hkv, 64)
    ref = torch.empty(plan.total_rows, d, dtype=torch.float64)
    for hh in range(hq):
        sc = qs[:, hh] @ kk[:, hh // (hq // hkv)].T * 0.6931471805599453
        ref[:, hh * 64:(hh + 1) * 64] = torch.softmax(sc, -1) @ vv[:, hh // (hq // hkv)]
    assert_close(out.float(), ref, "bf16", scale=2.0)


@pytest.mark.parametrize("case", [([(4, 16, 16)], [1]), ([(8, 32, 48), (4, 8, 24), (16, 64, 64)], [5, 3, 128]),
                                  ([(16, 128, 128)], [128]), ([(4, 8, 8), (4, 8, 8)], [0, 63]), ([(16, 64, 64)] * 3, [0, 61, 128])])
@pytest.mark.parametrize("heads", [(4, 2), (12, 4), (2, 2)])
def test_attention64_varlen_gqa_gate(case, heads):
    """ttv_attention64 (64 query rows per wave, one wave per SIMD) against the oracle's per-sequence softmax attention
    (reference transformer.py:100,103): ragged sequences (lengths 5 .. 1152, not multiples of 64: clamped rows, masked keys, idle
    waves), GQA groups of 1, 2 and 3 q-heads, with and without the gate."""
    shapes, counts = case
    plan = BatchPlan(shapes, counts, (4, 8, 8), DEV)
    hq, hkv = heads
    d, gq = hq * 64, hkv * 64
    ld = 2 * d + 2 * gq
    g = torch.Generator().manual_seed(len(shapes) + hq)
    qkvg = torch.randn(plan.total_rows, ld, generator=g)
    qkvg[:, :d] *= 2.0
    q_f32 = qkvg[:, :d].clone()
    qkvg = qkvg.to(torch.bfloat16)
    out = torch.empty(plan.total_rows, d, dtype=torch.bfloat16, device=DEV)
    xd = qkvg.to(DEV)
    xd[:, :d] = (q_f32 * (0.125 * 1.4426950408889634)).to(torch.bfloat16).to(DEV)   # TTV_ATTN_QSCALED
    tab = plan.attention_table64(hq, hkv)
    f = qkvg.float()
    q, gt, k, v = f.split([d, d, gq, gq], dim=-1)
    ref0 = O.attention_varlen(q.unflatten(-1, (hq, 64)), k.unflatten(-1, (hkv, 64)), v.unflatten(-1, (hkv, 64)), plan.cu_seqlens).flatten(-2)
    for gate in (1, 0):
        out.fill_(float("nan"))
        _lib.check(L().ttv_attention64(xd.data_ptr(), ld, out.data_ptr(), d, plan.cu_dev.data_ptr(), tab.data_ptr(), tab.shape[0], hq, hkv, 64,
                                       gate | 4, _lib.TTV_BF16, S()), "attention64")
        ref = ref0 * torch.sigmoid(gt) if gate else ref0
        assert_close(out.float(), ref, "bf16", scale=2.0)


@pytest.mark.parametrize("spike", [6.0, 30.0])
@pytest.mark.parametrize("spike_key", [200, 204, 20, 264])
def test_attention64_reference_shift_branch(spike, spike_key):
    """Force the softmax reference of ttv_attention64 to move at a chosen key tile (one key spiked against one query direction): the
    wave-uniform shift of scores, row sum, O (accumulation registers) and the -m start vector, for tile A and tile B rows, in the
    first tile, a middle tile and the masked last tile."""
    plan = BatchPlan([(16, 64, 64)], [9], (4, 8, 8), DEV)  # S = 265 -> 5 key tiles, the last with 9 keys
    hq, hkv, d, gq = 4, 2, 256, 128
    ld = 2 * d + 2 * gq
    g = torch.Generator().manual_seed(3)
    x = torch.randn(plan.total_rows, ld, generator=g) * 0.5
    q = x[:, :d].view(-1, 4, 64)
    x[spike_key, 2 * d: 2 * d + gq] = spike * torch.sign(q[5, 0]).repeat(2)   # dominates for query 5 (tile A) and its neighbours
    x[40, :64] = q[5, 0]                                                      # the same direction in a tile-B row (row 40 of its wave)
    q_f32 = x[:, :d].clone()
    x = x.to(torch.bfloat16)
    out = torch.empty(plan.total_rows, d, dtype=torch.bfloat16, device=DEV)
    xd = x.to(DEV)
    xd[:, :d] = (q_f32 * (0.125 * 1.4426950408889634)).to(torch.bfloat16).to(DEV)
    tab = plan.attention_table64(hq, hkv)
    _lib.check(L().ttv_attention64(xd.data_ptr(), ld, out.data_ptr(), d, plan.cu_dev.data_ptr(), tab.data_ptr(), tab.shape[0], hq, hkv, 64, 4,
                                   _lib.TTV_BF16, S()), "attention64")
    f = x.float()
    qq, gt, k, v = f.split([d, d, gq, gq], dim=-1)
    ref = O.attention_varlen(qq.unflatten(-1, (hq, 64)), k.unflatten(-1, (hkv, 64)), v.unflatten(-1, (hkv, 64)), plan.cu_seqlens).flatten(-2)
    assert_close(out.float(), ref, "bf16", scale=2.0)


# ---------------------------------------------------------------------------------------------- patches
@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_patch_gather_scatter(dt):
    shapes = [(8, 16, 24), (4, 8, 8), (16, 32, 16)]
    plan = BatchPlan(shapes, [1, 1, 1], (4, 8, 8), DEV)
    g = torch.Generator().manual_seed(2)
    clips = [torch.randn(3, *s, generator=g).to(DT[dt]) for s in shapes]
    cd = [c.to(DEV) for c in clips]
    pd = 768
    patches = torch.empty(plan.sum_patches, pd, dtype=DT[dt], device=DEV)
    code = _lib.dtype_code(DT[dt])
    _lib.check(L().ttv_patch_gather(_lib.ptr_array(cd), plan.clip_desc_dev.data_ptr(), 0, 3, 4, 8, 8, 3, patches.data_ptr(), pd, code,
                                    max(plan.grid_sizes), S()), "gather")
    perm = torch.arange(768).reshape(4, 8, 8, 3).permute(3, 0, 1, 2).reshape(-1)   # (c,pt,ph,pw) <- (pt,ph,pw,c)
    ref = torch.cat([O.patchify(c, (4, 8, 8)) for c in clips], 0)[:, perm]
    assert torch.equal(patches.cpu(), ref)        # pure data movement: bit-exact
    outs = [torch.zeros_like(c) for c in cd]
    _lib.check(L().ttv_patch_scatter(patches.data_ptr(), pd, plan.clip_desc_dev.data_ptr(), 0, 3, 4, 8, 8, 3, _lib.ptr_array(outs), code,
                                     max(plan.grid_sizes), S()), "scatter")
    for o, c in zip(outs, clips):
        assert torch.equal(o.cpu(), c)


def test_patch_kat_matches_reference_fixture():
    d = np.load(os.path.join(G, "patch_kat.npz"))
    clip = torch.from_numpy(d["clip"]).to(DEV)
    plan = BatchPlan([tuple(clip.shape[1:])], [1], (4, 8, 8), DEV)
    patches = torch.empty(plan.sum_patches, 768, dtype=torch.float32, device=DEV)
    _lib.check(L().ttv_patch_gather(_lib.ptr_array([clip]), plan.clip_desc_dev.data_ptr(), 0, 1, 4, 8, 8, 3, patches.data_ptr(), 768,
                                    _lib.TTV_F32, plan.sum_patches, S()), "gather")
    inv = torch.empty(768, dtype=torch.long)
    inv[torch.arange(768).reshape(4, 8, 8, 3).permute(3, 0, 1, 2).reshape(-1)] = torch.arange(768)
    assert np.array_equal(patches.cpu()[:, inv].numpy(), d["patches"])


# ---------------------------------------------------------------------------------------------- histogram
def test_codebook_histogram():
    g = torch.Generator().manual_seed(4)
    idx = torch.randint(0, 4375, (100000,), generator=g, dtype=torch.int32)
    counts = torch.zeros(4375, dtype=torch.int64, device=DEV)
    idd = idx.to(DEV)
    _lib.check(L().ttv_codebook_histogram(idd.data_ptr(), idx.numel(), counts.data_ptr(), 4375, S()), "hist")
    assert torch.equal(counts.cpu(), torch.bincount(idx.long(), minlength=4375))


def test_errors_are_loud():
    with pytest.raises(RuntimeError, match="HIP"):
        FSQ([7, 5, 5, 5, 5])(torch.zeros(3, 5))            # CPU tensor: no fallback
    x = torch.zeros(4, 100, device=DEV)
    with pytest.raises(RuntimeError, match="K="):
        _lib.check(L().ttv_linear(x.data_ptr(), 100, x.data_ptr(), 100, None, None, x.data_ptr(), 4, 4, 4, 98, _lib.TTV_F32, S()), "linear")


# ---------------------------------------------------------------------------------------------- PSNR statistic
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_psnr_matches_the_reference_definition(dtype):
    """EvalMetrics (reference model/metrics/eval_metrics.py:19,32-36): x.clamp(-1, 1), PSNR with data_range 2 over ALL elements of all
    updates = 10 log10(4 / MSE); fused squared-error reduction against float64 on the host, ragged clip shapes, two updates."""
    from titok_video_amd.model.metrics.eval_metrics import EvalMetrics
    g = torch.Generator().manual_seed(4)
    shapes = [(3, 4, 16, 24), (3, 8, 8, 8), (3, 4, 40, 8), (3, 2, 5, 7)]
    target = [(torch.rand(s, generator=g) * 2 - 1).to(dtype) for s in shapes]
    recon = [(t.float() + 0.3 * torch.randn(t.shape, generator=g)).to(dtype) for t in target]      # some values leave [-1, 1]: clamped
    m = EvalMetrics()
    m.update([r.to(DEV) for r in recon[:3]], [t.to(DEV) for t in target[:3]])
    m.update([recon[3].to(DEV)], [target[3].to(DEV)])
    got = m.compute()["eval/psnr"]
    sq = sum(float(((r.double().clamp(-1, 1) - t.double()) ** 2).sum()) for r, t in zip(recon, target))
    n = sum(t.numel() for t in target)
    ref = 10.0 * np.log10(4.0 * n / sq)
    assert abs(got - ref) < (1e-4 if dtype == torch.float32 else 1e-3), (got, ref)
    m.reset()
    m.update([target[0].to(DEV)], [target[0].to(DEV)])
    assert m.compute()["eval/psnr"] == float("inf")


# ---------------------------------------------------------------------------------------------- loader tail
@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("shape", [(4, 16, 16), (8, 24, 40), (16, 168, 168)])
def test_clip_from_u8_equals_the_host_normalisation(dt, shape):
    """ttv_clip_from_u8 (uint8 [T,H,W,3] -> [3,T,H,W], u8 / 127.5 - 1, reference video_dataset.py:116-119) bit-equal to the torch
    expression the host-side reader evaluates (shards.shard_samples), every byte value, both dtypes."""
    t, h, w = shape
    g = torch.Generator().manual_seed(t * h)
    frames = torch.randint(0, 256, (t, h, w, 3), generator=g, dtype=torch.uint8)
    frames.view(-1)[:256] = torch.arange(256, dtype=torch.uint8)
    ref = (frames.permute(3, 0, 1, 2).to(torch.float32) / 127.5 - 1.0).to(DT[dt]).contiguous()
    d8 = frames.to(DEV)
    out = torch.empty((3, t, h, w), dtype=DT[dt], device=DEV)
    _lib.check(L().ttv_clip_from_u8(d8.data_ptr(), t, h, w, out.data_ptr(), _lib.dtype_code(DT[dt]), S()), "clip_from_u8")
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("shape", [(300, 256, 768), (1000, 1408, 256), (129, 768, 704), (36864, 256, 256)])
def test_linear_f32_lds_dma_staging_is_bit_identical(shape):
    """Round 4: the exact-fp32 GEMM stages both operands by LDS-DMA when K is a multiple of 32 (k_gemm_split_dma<.., false>); ttv_debug_set
    bit 13 selects the register-staged kernel.  Same products, same summation order: equal bit for bit (and against float64 as usual)."""
    M, N, K = shape
    x, w, g = _lin_inputs(M, N, K, "f32", M + K)
    b = (torch.randn(N, generator=g) * 0.1)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    outs = []
    for bit in (0, 8192):
        y = torch.full((M, N), float("nan"), device=DEV)
        L().ttv_debug_set(bit)
        try:
            _lib.check(L().ttv_linear(xd.data_ptr(), K, wd.data_ptr(), K, bd.data_ptr(), None, y.data_ptr(), N, M, N, K, _lib.TTV_F32, S()), "linear")
            torch.cuda.synchronize()
        finally:
            L().ttv_debug_set(0)
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    assert_close(outs[0], x.double() @ w.double().T + b.double(), "f32")
