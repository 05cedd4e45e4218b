"""End-to-end parity of the HIP towers / TiTok facade against fixtures produced by the reference's own code
(tests/golden/*.npz) and against the CPU oracle on the same seeded inputs.  `-m gpu`.

Parity bars
  * fp32 compute  : token indices BIT-EXACT vs the reference on every fixture token whose FSQ rounding margin
                    0.5-|b-round(b)| exceeds 1e-3 (all tokens of the small fixtures); z / pixels to ~1e-3.
  * bf16 compute  : (the configuration the benchmark runs) indices exact on tokens with margin > TAU_BF16, raw match
                    rate reported and bounded below; pixels within PIX_TOL_BF16 (max-abs on a [-1,1]-scaled signal with
                    std ~1.9) and 2.5% relative Frobenius error.  SURVEY.md R8: the reference's own bf16 vs fp32 runs
                    agree on only 94.9% of indices, so bf16 can not be bit-exact by construction.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import titok_oracle as O
from titok_video_amd.model.base.blocks import TiTokEncoder
from titok_video_amd.model.titok import TiTok
from titok_video_amd.synthetic import seeded_titok_state, seeded_tower_state, synthetic_clips

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
LEVELS = [7, 5, 5, 5, 5]
TAU_F32 = 1e-3
TAU_BF16 = 0.08
PIX_TOL_BF16 = 0.25


def config(levels=LEVELS, enc="tiny", dec="tiny"):
    return SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(
        patch_size=[4, 8, 8], fsq_levels=list(levels), encoder_size=enc, decoder_size=dec)))


def build(dtype, seed=0, **kw):
    m = TiTok(config(**kw))
    m.load_state_dict(seeded_titok_state(seed, encoder_size=kw.get("enc", "tiny"), decoder_size=kw.get("dec", "tiny")), strict=True)
    return m.to(DEV, dtype).eval()


def fixture_inputs(d, dtype):
    if "shapes" in d:
        shapes, counts = d["shapes"].tolist(), d["counts"].tolist()
    else:
        shapes, counts = [d["shape"].tolist()], [int(d["count"])]
    clips = synthetic_clips(shapes, seed=int(d["clip_seed"]), dtype=dtype, device=DEV)
    return shapes, counts, clips


def run(model, clips, counts):
    with torch.no_grad():
        codes, dd = model.encode(clips, counts, want_bounded=True)
        recon = model.decode(codes, counts, [tuple(c.shape[1:]) for c in clips])
    torch.cuda.synchronize()
    return codes, dd["indices"].cpu().numpy(), model.last_bounded.cpu(), recon


@pytest.mark.parametrize("name", ["titok_small.npz", "titok_single.npz"])
def test_fp32_bit_exact_indices_vs_reference(name):
    d = np.load(os.path.join(G, name))
    model = build(torch.float32)
    shapes, counts, clips = fixture_inputs(d, torch.float32)
    codes, idx, bounded, recon = run(model, clips, counts)
    assert np.array_equal(idx, d["indices"])                     # every token, bit-exact
    np.testing.assert_allclose(bounded.numpy(), d["bounded"], rtol=0, atol=2e-3)
    refs = [d[f"recon_{i}"] for i in range(len(recon))] if "recon_0" in d else [d["recon"]]
    for r, ref in zip(recon, refs):
        np.testing.assert_allclose(r.cpu().numpy(), ref, rtol=0, atol=5e-3)


def test_fp32_cfg1_vs_reference():
    """BASELINE config #1 shapes (4 x 16x128x128, K=128)."""
    d = np.load(os.path.join(G, "titok_cfg1.npz"))
    model = build(torch.float32)
    shapes, counts, clips = fixture_inputs(d, torch.float32)
    codes, idx, bounded, recon = run(model, clips, counts)
    margin = O.fsq_margin(torch.from_numpy(d["bounded"])).numpy()
    safe = margin > TAU_F32
    assert np.array_equal(idx[safe], d["indices"][safe])
    raw = float((idx == d["indices"]).mean())
    print(f"cfg1 fp32: raw index match {raw:.4f}, tokens with margin>{TAU_F32}: {safe.mean():.4f}")
    assert raw > 0.99
    rs = torch.stack([r.float().cpu() for r in recon])
    np.testing.assert_allclose(rs[:, :, ::4, ::8, ::8].numpy(), d["recon_sample"], rtol=0, atol=2e-2)
    assert abs(float(rs.double().std()) - float(d["recon_std"])) < 1e-3


@pytest.mark.parametrize("name", ["titok_small.npz", "titok_cfg1.npz"])
def test_bf16_vs_reference(name):
    """bf16 compute (the benchmark configuration).  Yardstick: the reference's OWN modules run in bf16 on the same
    inputs/weights (fixture keys *_refbf16) - the HIP bf16 path must be at least as close to the reference's fp32
    result as the reference's own bf16 execution is."""
    d = np.load(os.path.join(G, name))
    model = build(torch.bfloat16)
    shapes, counts, clips = fixture_inputs(d, torch.bfloat16)
    codes, idx, bounded, recon = run(model, clips, counts)
    ref_b = torch.from_numpy(d["bounded"])
    margin = O.fsq_margin(ref_b).numpy()
    raw = float((idx == d["indices"]).mean())
    raw_ref16 = float((d["indices_refbf16"] == d["indices"]).mean())
    err = (bounded - ref_b).abs()
    err_ref16 = (torch.from_numpy(d["bounded_refbf16"]) - ref_b).abs()
    print(f"{name} bf16: index match vs fp32 reference: HIP {raw:.4f} | reference-in-bf16 {raw_ref16:.4f}; "
          f"mean|bounded err| HIP {float(err.mean()):.4f} | ref-bf16 {float(err_ref16.mean()):.4f}; "
          f"max HIP {float(err.max()):.4f} | ref-bf16 {float(err_ref16.max()):.4f}")
    # raw agreement is a count of coin flips near rounding boundaries: allow 3 % or one token and a half on the small fixtures
    # (17 tokens); the error of the continuous value underneath is the real bar
    assert raw >= raw_ref16 - max(0.03, 1.5 / idx.size)
    assert float(err.mean()) <= 1.15 * float(err_ref16.mean())
    # exact wherever the fp32 value is further from a rounding boundary than the observed bf16 error
    safe = margin > float(err.max()) + 1e-6
    assert np.array_equal(idx[safe], d["indices"][safe])
    # decoder alone on the reference's fp32-run codes (independent of index flips): pixel error no worse than the
    # reference's own bf16 decoder (x1.15) and < PIX_TOL_BF16 absolute / 2.5% relative
    ref_codes = O.fsq_indices_to_codes(torch.from_numpy(d["indices"]), LEVELS).to(torch.bfloat16).to(DEV)
    with torch.no_grad():
        rec2 = model.decode(ref_codes, counts, shapes)
    if "recon_0" in d:
        mine = torch.cat([r.float().cpu().flatten() for r in rec2])
        ref = torch.cat([torch.from_numpy(d[f"recon_{i}"]).flatten() for i in range(len(rec2))])
        ref16 = torch.cat([torch.from_numpy(d[f"recon_refbf16_{i}"]).flatten() for i in range(len(rec2))])
    else:
        mine = torch.stack([r.float().cpu() for r in rec2])[:, :, ::4, ::8, ::8].flatten()
        ref = torch.from_numpy(d["recon_sample"]).flatten()
        ref16 = torch.from_numpy(d["recon_sample_refbf16"]).flatten()
    e_mine, e_ref16 = float((mine - ref).norm() / ref.norm()), float((ref16 - ref).norm() / ref.norm())
    print(f"   decoder rel. error vs fp32 reference: HIP {e_mine:.5f} | reference-in-bf16 {e_ref16:.5f}; max abs HIP {float((mine - ref).abs().max()):.4f}")
    assert e_mine <= 1.15 * e_ref16 and e_mine < 0.025
    assert float((mine - ref).abs().max()) < PIX_TOL_BF16


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_api_contract(dtype):
    """Shapes / dtypes / list-of-clips API of model/titok.py:47-74; decode_indices == decode."""
    model = build(dtype)
    shapes = [(4, 16, 16), (8, 32, 48), (4, 8, 24)]
    counts = [2, 5, 3]
    clips = synthetic_clips(shapes, seed=5, dtype=dtype, device=DEV)
    tc = torch.tensor(counts, dtype=torch.int32, device=DEV)       # the reference passes a device tensor
    with torch.no_grad():
        recon, out = model(clips, tc)
        assert out["indices"].dtype == torch.int32 and out["indices"].shape == (10,)
        assert all(r.shape == c.shape and r.dtype == dtype and r.device == c.device for r, c in zip(recon, clips))
        x_q, od = model.encode(clips, tc, grids=torch.tensor(shapes, dtype=torch.int32, device=DEV))
        assert x_q.shape == (10, 5) and x_q.dtype == dtype
        assert torch.equal(od["indices"], out["indices"])
        a = model.decode(x_q, tc, shapes)
        b = model.decode_indices(od["indices"], shapes, tc)
        c = model.decode_indices(list(torch.split(od["indices"], counts)), shapes)
        for u, v, w, r in zip(a, b, c, recon):
            assert torch.equal(u, v) and torch.equal(u, w) and torch.equal(u, r)
        _, sp = model.encode(clips, counts, split_indices=True)
        assert [int(t.shape[0]) for t in sp["indices"]] == counts
        # FSQ consistency: indices_to_codes(indices) == codes
        assert torch.equal(model.quantize.indices_to_codes(od["indices"], dtype=dtype), x_q)
        assert int(od["indices"].min()) >= 0 and int(od["indices"].max()) < model.quantize.codebook_size


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_packing_invariance(dtype):
    """A clip's tokens and pixels must not depend on its batch mates (block-diagonal attention, transformer.py:100)."""
    model = build(dtype)
    shapes = [(8, 32, 48), (4, 16, 16), (16, 64, 32)]
    counts = [5, 1, 17]
    clips = synthetic_clips(shapes, seed=11, dtype=dtype, device=DEV)
    with torch.no_grad():
        recon, out = model(clips, counts)
        for i in range(3):
            r1, o1 = model([clips[i]], [counts[i]])
            s = sum(counts[:i])
            assert torch.equal(o1["indices"], out["indices"][s:s + counts[i]])
            assert torch.equal(r1[0], recon[i])


def test_more_clips_than_one_pointer_table_bf16():
    """Batches of more than TTV_MAX_CLIPS_PER_LAUNCH (64) clips take the stand-alone patch copy / ln_post kernels instead of
    the GEMM-fused gather / scatter: same results up to bf16 rounding of the folded gain."""
    model = build(torch.bfloat16)
    n = 70
    shapes = [(4, 16, 16)] * n
    counts = [2] * n
    clips = synthetic_clips(shapes, seed=5, dtype=torch.bfloat16, device=DEV)
    with torch.no_grad():
        recon, out = model(clips, counts)                       # 70 clips: unfused patch paths
        ra, oa = model(clips[:35], counts[:35])                 # <= 64 clips: fused paths
        rb, ob = model(clips[35:], counts[35:])
    idx2 = torch.cat([oa["indices"], ob["indices"]])
    assert torch.equal(out["indices"], idx2)                    # the encoder side differs only by where the operand is read from
    worst = max(float((x.float() - y.float()).abs().max()) for x, y in zip(recon, list(ra) + list(rb)))
    assert worst < 0.08, worst


def test_discriminator_style_encoder_call():
    """ReconstructionLoss builds TiTokEncoder(out_channels=1) and calls it with K=4 register tokens (loss_module.py:43-48,96-101)."""
    enc = TiTokEncoder(model_size="tiny", patch_size=(4, 8, 8), in_channels=3, out_channels=1)
    sd = seeded_tower_state("encoder", "tiny", (4, 8, 8), 3, 1, seed=77)
    enc.load_state_dict(sd, strict=True)
    enc = enc.to(DEV, torch.float32)
    shapes = [(8, 16, 16), (4, 24, 16)]
    clips = synthetic_clips(shapes, seed=3, dtype=torch.float32, device=DEV)
    tc = torch.tensor([4, 4], dtype=torch.int32, device=DEV)
    with torch.no_grad():
        logits = enc(clips, tc)
    assert logits.shape == (8, 1)
    ref = O.encoder_forward([c.cpu() for c in clips], [4, 4], sd, "tiny", (4, 8, 8))
    np.testing.assert_allclose(logits.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-3)


def test_blocks_kat_fp32():
    """Transformer stack alone vs the reference fixture (Attn + GEGLU + KEEL layers), through the encoder's kernels."""
    import ctypes as C
    from titok_video_amd import _lib
    from titok_video_amd.plan import BatchPlan
    d = np.load(os.path.join(G, "blocks_kat.npz"))
    sd = seeded_titok_state(int(d["weight_seed"]))
    grids, counts = d["grids"].tolist(), d["counts"].tolist()
    plan = BatchPlan([(g[0] * 4, g[1] * 8, g[2] * 8) for g in grids], counts, (4, 8, 8), DEV)
    x = torch.from_numpy(d["x"]).to(DEV)
    Lr, dm, gq, I = x.shape[0], 256, 128, 704
    S = _lib.stream_ptr(torch.device(DEV))
    lib = _lib.lib()
    p = "encoder.model_layers."
    w = {k: v.to(DEV) for k, v in sd.items() if k.startswith(p)}
    # attention sub-layer 1
    xn = torch.empty_like(x)
    _lib.check(lib.ttv_rmsnorm(x.data_ptr(), 1, dm, None, xn.data_ptr(), 1, dm, None, w[p + "attn_layer.1.pre_ln.weight"].data_ptr(), Lr, dm, 1e-5, S), "n")
    qkv = torch.empty(Lr, 2 * dm + 2 * gq, device=DEV)
    _lib.check(lib.ttv_linear_qkv_rope(xn.data_ptr(), dm, w[p + "attn_layer.1.to_qkv.weight"].data_ptr(), dm, qkv.data_ptr(), 2 * dm + 2 * gq, Lr, dm, gq, plan.rope_cs.data_ptr(), 1, S), "q")
    ao = torch.empty(Lr, dm, device=DEV)
    _lib.check(lib.ttv_attention(qkv.data_ptr(), 2 * dm + 2 * gq, ao.data_ptr(), dm, plan.cu_dev.data_ptr(), plan.attention_table(4, 2).data_ptr(), plan.attention_table(4, 2).shape[0], 4, 2, 64, 1, 1, S), "a")
    out = torch.empty(Lr, dm, device=DEV)
    _lib.check(lib.ttv_linear(ao.data_ptr(), dm, w[p + "attn_layer.1.out_proj.weight"].data_ptr(), dm, None, None, out.data_ptr(), dm, Lr, dm, dm, 1, S), "o")
    np.testing.assert_allclose(out.cpu().numpy(), d["attn1"], rtol=1e-3, atol=1e-3)
    # GEGLU sub-layer 1
    _lib.check(lib.ttv_rmsnorm(x.data_ptr(), 1, dm, None, xn.data_ptr(), 1, dm, None, w[p + "ffd_layer.1.norm.weight"].data_ptr(), Lr, dm, 1e-5, S), "n")
    h = torch.empty(Lr, I, device=DEV)
    _lib.check(lib.ttv_linear_geglu(xn.data_ptr(), dm, w[p + "ffd_layer.1.w12.weight"].data_ptr(), dm, h.data_ptr(), I, Lr, I, dm, 1, S), "g")
    _lib.check(lib.ttv_linear(h.data_ptr(), I, w[p + "ffd_layer.1.w3.weight"].data_ptr(), I, None, None, out.data_ptr(), dm, Lr, dm, I, 1, S), "o")
    np.testing.assert_allclose(out.cpu().numpy(), d["ffd1"], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("size", ["small", "base", "large"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_other_model_sizes_match_oracle(size, dtype):
    """get_model_dims sizes beyond tiny (utils.py:8-23): widths 512/768/1024, GQA 8:2 / 12:4 / 16:4, GEGLU inner 1376 /
    2048 / 2752 (1376 is not a multiple of the 64-wide feature tiles) - generic (unfused) kernel path vs the CPU oracle."""
    sd = seeded_titok_state(3, encoder_size=size, decoder_size=size, gain=3.0)
    m = TiTok(config(enc=size, dec=size))
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV, dtype).eval()
    shapes, counts = [(4, 16, 16), (8, 16, 24)], [3, 6]
    clips_cpu = synthetic_clips(shapes, seed=13)
    with torch.no_grad():
        ref_recon, ref_idx, ref_z, ref_b = O.titok_forward(clips_cpu, counts, sd, LEVELS, size, size)
        codes, dd = m.encode([c.to(DEV, dtype) for c in clips_cpu], counts, want_bounded=True)
        recon = m.decode(O.fsq_indices_to_codes(ref_idx, LEVELS).to(DEV, dtype), counts, shapes)   # decoder on the oracle's codes
    berr = float((m.last_bounded.cpu() - ref_b).abs().max())
    perr = max(float((r.float().cpu() - rr).abs().max()) for r, rr in zip(recon, ref_recon))
    scale = max(float(rr.abs().max()) for rr in ref_recon)
    print(f"{size} {dtype}: max |bounded err| {berr:.2e}, max pixel err {perr:.2e} (pixel scale {scale:.2f})")
    if dtype == torch.float32:
        assert torch.equal(dd["indices"].cpu(), ref_idx)
        assert berr < 2e-3 and perr < 5e-3 * max(1.0, scale)
    else:
        safe = O.fsq_margin(ref_b) > berr + 1e-6
        assert torch.equal(dd["indices"].cpu()[safe], ref_idx[safe])
        assert berr < 0.6 and perr < 0.08 * max(1.0, scale)


def test_ragged_dynamic_batches_fp32_match_oracle():
    """Batches as the reference's loader emits them (token-budget packing, ragged shapes, K ~ U[1,128], device int32
    token_counts): indices bit-exact vs the oracle, checkpoint-style state dict with the trainer's `model.` prefix."""
    from titok_video_amd.data import SyntheticClipStream, dynamic_batches
    sd = seeded_titok_state(0)
    lightning_style = {"model." + k: v for k, v in sd.items()}
    m = TiTok(config())
    m.load_state_dict({k[len("model."):]: v for k, v in lightning_style.items() if k.startswith("model.")}, strict=True)
    m = m.to(DEV, torch.float32).eval()
    stream = SyntheticClipStream(min_grid=(4, 32, 32), max_grid=(8, 64, 64), dtype=torch.float32, device=DEV, seed=2, length=7)
    n = 0
    for batch in dynamic_batches(stream, (4, 8, 8), (1, 16), 200, seed=4, device=DEV):
        with torch.no_grad():
            recon, out = m(batch["video"], batch["token_counts"])
        counts = batch["token_counts"].tolist()
        ref_recon, ref_idx, _, _ = O.titok_forward([v.cpu() for v in batch["video"]], counts, sd, LEVELS)
        assert torch.equal(out["indices"].cpu(), ref_idx)
        for r, rr in zip(recon, ref_recon):
            assert float((r.cpu() - rr).abs().max()) < 5e-3
        n += len(counts)
    assert n == 7


def test_forward_pipeline_matches_sequential_calls():
    """Two batches in flight on two streams (titok_video_amd.pipeline): bit-identical to one batch at a time, in order."""
    from titok_video_amd.pipeline import ForwardPipeline
    model = build(torch.bfloat16)
    batches = []
    for i, (shapes, counts) in enumerate([([(4, 16, 16), (8, 32, 48)], [3, 7]), ([(16, 64, 64)], [32]),
                                          ([(4, 8, 24), (4, 16, 16), (8, 32, 32)], [1, 2, 9]), ([(8, 32, 48)], [5])]):
        batches.append((synthetic_clips(shapes, seed=50 + i, dtype=torch.bfloat16, device=DEV), counts))
    with torch.no_grad():
        ref = [model(c, k) for c, k in batches]
    torch.cuda.synchronize()
    pipe = ForwardPipeline(model, depth=2)
    tickets = [pipe.submit(c, k) for c, k in batches]
    outs = [pipe.result(t) for t in tickets]
    torch.cuda.synchronize()
    for (r0, o0), (r1, o1) in zip(ref, outs):
        assert torch.equal(o0["indices"], o1["indices"])
        for a, b in zip(r0, r1):
            assert torch.equal(a, b)
    # the two streams really use separate scratch buffers
    assert len({k[2] for k in model.encoder._ws}) >= 2


def test_sampling_range_extremes_fp32_match_oracle():
    """The corners of the reference's sampling ranges (configs/tiny.yaml:57-62): the largest grid 16x168x168 (21x21 patches per
    frame group: not a power of two, 1764 patches) with the most (128) and the fewest (1) latent tokens, next to the smallest
    grid 8x128x128.  fp32 path against the oracle: indices bit-exact away from rounding boundaries, pixels to 5e-3."""
    shapes, counts = [(16, 168, 168), (8, 128, 128), (16, 168, 168)], [128, 1, 1]
    clips = synthetic_clips(shapes, seed=77, dtype=torch.float32, device=DEV)
    model = build(torch.float32)
    with torch.no_grad():
        codes, od = model.encode(clips, counts, want_bounded=True)
        recon = model.decode(codes, counts, shapes)
    sd = seeded_titok_state(0)
    ref_recon, ref_idx, _ref_z, ref_bounded = O.titok_forward([c.cpu() for c in clips], counts, sd, LEVELS)
    idx = od["indices"].cpu()
    assert idx.shape == (130,)
    safe = O.fsq_margin(ref_bounded) > TAU_F32
    assert int(safe.sum()) >= 120
    assert torch.equal(idx[safe], ref_idx[safe])
    np.testing.assert_allclose(model.last_bounded.cpu().numpy(), ref_bounded.numpy(), rtol=0, atol=2e-3)
    for r, ref in zip(recon, ref_recon):
        assert r.shape == ref.shape
        # decoder inputs are the same codes wherever the indices agree; compare through the oracle's decode of OUR indices
    dec_ref = O.titok_decode_indices(idx, shapes, counts, sd, LEVELS)
    for r, ref in zip(recon, dec_ref):
        np.testing.assert_allclose(r.cpu().numpy(), ref.numpy(), rtol=0, atol=5e-3)


def test_sampling_range_extremes_bf16_close_to_oracle():
    """Same corners through the bf16 kernels (fused tail, gathered proj_in, scattered proj_out, LDS-DMA attention with an odd
    number of key tiles): indices exact wherever the fp32 value is further from a rounding boundary than the observed error."""
    shapes, counts = [(16, 168, 168), (8, 128, 128), (16, 168, 168)], [128, 1, 1]
    clips32 = synthetic_clips(shapes, seed=77, dtype=torch.float32, device="cpu")
    model = build(torch.bfloat16)
    clips = [c.to(DEV, torch.bfloat16) for c in clips32]
    with torch.no_grad():
        codes, od = model.encode(clips, counts, want_bounded=True)
        recon = model.decode(codes, counts, shapes)
    sd = seeded_titok_state(0)
    _r, ref_idx, _z, ref_bounded = O.titok_forward([c.to(torch.bfloat16).float() for c in clips32], counts, sd, LEVELS)
    berr = float((model.last_bounded.float().cpu() - ref_bounded).abs().max())
    safe = O.fsq_margin(ref_bounded) > berr + 1e-6
    assert berr < 0.6
    assert torch.equal(od["indices"].cpu()[safe], ref_idx[safe])
    dec_ref = O.titok_decode_indices(od["indices"].cpu(), shapes, counts, sd, LEVELS)
    for r, ref in zip(recon, dec_ref):
        assert float((r.float().cpu() - ref).abs().max()) < PIX_TOL_BF16 * 1.5
