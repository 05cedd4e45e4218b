"""Split-bf16 ("three-pass") arithmetic of the index-exact encoder (round 4; `-m gpu`).

Every matrix product of an fp32 tower can run as three bf16 MFMA passes on operands split into hi + lo (hi = bf16(x), lo = bf16(x - hi)):
a b ~ ah bh + ah bl + al bh with fp32 accumulation.  What is dropped (al bl, and the remainder of the two-term split) is ~2^-17 relative
per product - against 2^-9 for plain bf16 operands and 2^-24 for fp32.  Stated tolerances:
  * one linear / one attention call: relative Frobenius error against float64 < 2e-5 (bf16 operands give ~3e-3, the exact-fp32 kernels ~1e-7);
  * the encoder of BASELINE config #2's fixture (tests/golden/titok_cfg1.npz, generated from the reference's modules): max |pre-rounding
    FSQ value error| < 1e-3 against the reference's fp32 run, hence every token whose rounding margin exceeds 1e-3 carries the reference's
    index (the CPU emulation of the same arithmetic, tests/probes/split_bf16_probe.py, gives 5.8e-4 / 512 of 512 tokens).
The reference has no such mode (it computes in the parameter dtype, titok.py:61): this is a build-defined fast path to the metric's
"token-index exact-match", judged against the reference's own fp32 outputs."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import titok_oracle as O
from titok_video_amd import _lib
from titok_video_amd.model.titok import TiTok
from titok_video_amd.plan import BatchPlan
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def L():
    return _lib.lib()


def S():
    return _lib.stream_ptr(torch.device(DEV))


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def split_image(w):
    img = torch.empty_like(w)
    _lib.check(L().ttv_split3_pack(w.data_ptr(), w.shape[1], img.data_ptr(), w.shape[1], w.shape[0], w.shape[1], S()), "pack")
    return img


def test_split3_pack_layout():
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(37, 64, generator=g) * torch.exp(3 * torch.randn(37, 1, generator=g))).to(DEV)
    img = split_image(w).cpu().view(torch.bfloat16).view(37, 16, 2, 4)          # [row][group of 4 k][hi | lo][4]
    wc = w.cpu().view(37, 16, 4)
    hi = wc.to(torch.bfloat16)
    lo = (wc - hi.float()).to(torch.bfloat16)
    assert torch.equal(img[:, :, 0], hi) and torch.equal(img[:, :, 1], lo)
    assert float((hi.float() + lo.float() - wc).abs().max() / wc.abs().max()) < 2 ** -16


@pytest.mark.parametrize("M,N,K", [(1000, 768, 256), (333, 256, 768), (36864, 256, 704), (129, 1408, 256), (64, 128, 100)])
def test_linear_split3_against_float64(M, N, K):
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g) * 0.1
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y = torch.full((M, N), float("nan"), device=DEV)
    _lib.check(L().ttv_linear_split3(xd.data_ptr(), K, split_image(wd).data_ptr(), K, bd.data_ptr(), y.data_ptr(), N, M, N, K, S()), "split3")
    ref = x.double() @ w.double().t() + b.double()
    e3 = rel(y, ref)
    e16 = rel(x.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t() + b.double(), ref)
    print(f"linear {M}x{N}x{K}: split3 {e3:.2e}, bf16 operands {e16:.2e}")
    assert e3 < 2e-5 and e3 < e16 / 50


@pytest.mark.parametrize("case", [([(4, 16, 16)], [1]), ([(8, 32, 48), (4, 8, 24), (16, 64, 64)], [5, 3, 128])])
@pytest.mark.parametrize("heads", [(4, 2), (12, 4)])
@pytest.mark.parametrize("split", [0, 1])
def test_attention_split3_against_the_fp32_oracle(case, heads, split):
    shapes, counts = case
    plan = BatchPlan(shapes, counts, (4, 8, 8), DEV)
    hq, hkv = heads
    d, gq = hq * 64, hkv * 64
    ld = 2 * d + 2 * gq
    g = torch.Generator().manual_seed(len(shapes) + hq)
    qkvg = torch.randn(plan.total_rows, ld, generator=g)
    qkvg[:, :d] *= 2.0
    xd = qkvg.to(DEV)
    q, gt, k, v = qkvg.split([d, d, gq, gq], dim=-1)
    ref = O.attention_varlen(q.unflatten(-1, (hq, 64)).double(), k.unflatten(-1, (hkv, 64)).double(), v.unflatten(-1, (hkv, 64)).double(),
                             plan.cu_seqlens).flatten(-2)
    tab = plan.attention_table(hq, hkv, split)
    for gate in (1, 0):
        out = torch.full((plan.total_rows, d), float("nan"), device=DEV)
        _lib.check(L().ttv_attention(xd.data_ptr(), ld, out.data_ptr(), d, plan.cu_dev.data_ptr(), tab.data_ptr(), tab.shape[0], hq, hkv, 64,
                                     gate | 32, _lib.TTV_F32, S()), "attention split3")
        want = ref * torch.sigmoid(gt.double()) if gate else ref
        e = rel(out, want)
        assert e < 2e-5, (gate, e)


def test_attention_split3_spiked_key():
    """the running maximum jumps late (a key that dominates one query): the rescale branch of the exact online softmax"""
    plan = BatchPlan([(16, 64, 64)], [9], (4, 8, 8), DEV)
    hq, hkv, d, gq = 4, 2, 256, 128
    ld = 2 * d + 2 * gq
    g = torch.Generator().manual_seed(3)
    x = torch.randn(plan.total_rows, ld, generator=g) * 0.5
    x[200, 2 * d: 2 * d + gq] = 30 * torch.sign(x[:, :d].view(-1, 4, 64)[5, 0]).repeat(2)
    out = torch.empty(plan.total_rows, d, device=DEV)
    tab = plan.attention_table(hq, hkv, 0)
    _lib.check(L().ttv_attention(x.to(DEV).data_ptr(), ld, out.data_ptr(), d, plan.cu_dev.data_ptr(), tab.data_ptr(), tab.shape[0], hq, hkv, 64, 32,
                                 _lib.TTV_F32, S()), "attention")
    q, gt, k, v = x.double().split([d, d, gq, gq], dim=-1)
    ref = O.attention_varlen(q.unflatten(-1, (hq, 64)), k.unflatten(-1, (hkv, 64)), v.unflatten(-1, (hkv, 64)), plan.cu_seqlens).flatten(-2)
    assert rel(out, ref) < 2e-5


def _cfg(levels, size="tiny"):
    return SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=levels, encoder_size=size, decoder_size=size)))


def test_split3_encoder_keeps_the_reference_fp32_indices_on_the_benchmark_fixture():
    """BASELINE config #2's fixture: clips 0-3 of the benchmark batch through `set_index_exact('split3')` (fp32 master weights, fp32
    clips, split-bf16 encoder, bf16 decoder) against the REFERENCE's fp32 run."""
    fix = np.load(os.path.join(GOLD, "titok_cfg1.npz"))
    levels = [7, 5, 5, 5, 5]
    sd = seeded_titok_state(0, gain=6.0)
    clips_cpu = synthetic_clips([(16, 128, 128)] * 4, seed=1234)
    counts = [128] * 4
    ref_idx, ref_b = torch.from_numpy(fix["indices"]), torch.from_numpy(fix["bounded"])
    out = {}
    for mode in ("fp32", "split3"):
        model = TiTok(_cfg(levels))
        model.load_state_dict(sd, strict=True)
        model = model.to(DEV, torch.float32).eval().set_index_exact(mode)
        clips = [c.to(DEV) for c in clips_cpu]
        with torch.no_grad():
            recon, info = model(clips, counts)
            model.encode(clips, counts, want_bounded=True)
        out[mode] = (info["indices"].cpu(), model.last_bounded.cpu(), recon)
        assert recon[0].dtype == torch.bfloat16
    idx, b, recon = out["split3"]
    err = (b - ref_b).abs()
    margin = O.fsq_margin(ref_b)
    safe = margin > 1e-3
    print(f"split3 encoder vs the reference's fp32 run: max |bounded err| {float(err.max()):.2e} (exact-fp32 kernels {float((out['fp32'][1] - ref_b).abs().max()):.2e}), "
          f"indices equal {int((idx == ref_idx).sum())}/{idx.numel()}, mismatches at margin > 1e-3: {int((idx[safe] != ref_idx[safe]).sum())}/{int(safe.sum())}")
    assert float(err.max()) < 1e-3
    assert torch.equal(idx[safe], ref_idx[safe])
    # the decoder ran in bf16 on the same indices in both modes: identical reconstructions wherever the codes are identical
    if torch.equal(idx, out["fp32"][0]):
        assert all(torch.equal(a, c) for a, c in zip(recon, out["fp32"][2]))


def test_split3_encoder_mixed_shapes_against_the_oracle():
    levels = [7, 5, 5, 5, 5]
    sd = seeded_titok_state(0)
    shapes, counts = [(8, 32, 48), (4, 16, 16), (16, 64, 64)], [17, 1, 128]
    clips_cpu = synthetic_clips(shapes, seed=42)
    with torch.no_grad():
        _r, ref_idx, _z, ref_b = O.titok_forward(clips_cpu, counts, sd, levels)
    model = TiTok(_cfg(levels))
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV, torch.float32).eval().set_index_exact("split3")
    with torch.no_grad():
        codes, info = model.encode([c.to(DEV) for c in clips_cpu], counts, want_bounded=True)
    err = float((model.last_bounded.cpu() - ref_b).abs().max())
    safe = O.fsq_margin(ref_b) > 1e-3
    assert err < 1e-3, err
    assert torch.equal(info["indices"].cpu()[safe], ref_idx[safe])
    # training through a split tower is refused (inference arithmetic)
    model.train()
    with pytest.raises(RuntimeError):
        model([c.to(DEV).requires_grad_(True) for c in clips_cpu], counts)


def test_split_images_written_by_the_producers_change_no_bit():
    """The split towers' norm kernels, attention epilogue and GEGLU epilogue write the next linear's operand as its split image (same bytes
    as the fp32 tensor); with ttv_debug_set bit 12 the activations stay fp32 and the GEMM's staging threads split them.  Same values,
    same split: the encoder's tokens must be equal bit for bit - also between the LDS-DMA split GEMM (k_gemm_split_dma, both operands images)
    and the register-staged one (bit 13)."""
    levels = [7, 5, 5, 5, 5]
    sd = seeded_titok_state(0)
    shapes, counts = [(8, 32, 48), (4, 16, 16), (16, 64, 64)], [17, 1, 128]
    clips = [c.to(DEV) for c in synthetic_clips(shapes, seed=42)]
    model = TiTok(_cfg(levels))
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV, torch.float32).eval().set_index_exact("split3")
    outs = []
    for bit in (0, 8192, 4096):      # default (images + LDS-DMA GEMM) | images, register-staged GEMM | fp32 activations, split in the GEMM
        L().ttv_debug_set(bit)
        try:
            with torch.no_grad():
                z = model.encoder.run(clips, counts, None, None, want_z=True)["z"].clone()
            torch.cuda.synchronize()
        finally:
            L().ttv_debug_set(0)
        outs.append(z)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
