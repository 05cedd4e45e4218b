"""End-to-end parity of the HIP towers / TiTok facade against fixtures produced by the reference's own code
(tests/golden/*.npz) and against the CPU oracle on the same seeded inputs.  `-m gpu`.

Parity bars
  * fp32 compute  : token indices BIT-EXACT vs the reference on every fixture token whose FSQ rounding margin
                    0.5-|b-round(b)| exceeds 1e-3 (all tokens of the small fixtures); z / pixels to ~1e-3.
  * bf16 compute  : (the configuration the benchmark runs) can not be bit-exact against an fp32 run (round() after tanh;
                    SURVEY.md R8: the reference's own bf16 run agrees with its fp32 run on 80-82 % of the fixture tokens).  The
                    bar is the YARDSTICK - a bf16 execution of the same model on the same inputs (fixture keys *_refbf16: the
                    reference's own modules in bf16; where no fixture exists, the oracle run in bf16): with FIXED thresholds
                    (TAU_LIST) the HIP path may not have more index mismatches than the yardstick (+ BF16_SLACK), its mean / max
                    pre-rounding error may not exceed the yardstick's by more than 15 % / 25 %, and indices must be exact
                    wherever the margin exceeds the yardstick's own maximum error (a constant of the fixture, not of this path).
                    Pixels within PIX_TOL_BF16 (max-abs on a [-1,1]-scaled signal with std ~1.9) and 2.5 % relative.
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
TAU_LIST = (0.08, 0.16)     # fixed rounding-margin thresholds of the bf16 mismatch counts (2x and 4x the mean bf16 error)
PIX_TOL_BF16 = 0.25


def bf16_slack(n_tokens, yard_mismatches=0):
    """index mismatches are coin flips near rounding boundaries: allow 1.5 % of the tokens, at least two - and not less than two
    standard deviations of the yardstick's own count (binomial: two bf16 executions that round differently - another softmax
    reference, another summation order - flip different marginal tokens; round 4 saw 99, 105 and 109 of 512 against a yardstick
    of 100 from three builds whose MEAN error agreed to 0.2 %).  The mean / maximum error ratios below are the sharp statistics."""
    sigma = (yard_mismatches * (1.0 - yard_mismatches / max(n_tokens, 1))) ** 0.5
    return max(2, int(round(0.015 * n_tokens)), int(np.ceil(2.0 * sigma)))


def assert_bf16_not_worse_than_yardstick(name, idx, bounded, ref_idx, ref_b, yard_idx, yard_b):
    """idx / bounded: HIP bf16 run; ref_*: fp32 reference; yard_*: a bf16 execution of the same model (reference modules or oracle).
    One set of thresholds for every caller (error ratios 1.15 / 1.25): tests whose batches held a few dozen tokens were given
    >= 128 tokens per clip in round 3 instead of a wider gate."""
    r_mean, r_max, extra = 1.15, 1.25, 0
    idx, ref_idx, yard_idx = (np.asarray(a) for a in (idx, ref_idx, yard_idx))
    margin = O.fsq_margin(ref_b).numpy()
    err, yerr = (bounded.float() - ref_b).abs(), (yard_b.float() - ref_b).abs()
    n = idx.size
    raw, yraw = int((idx != ref_idx).sum()), int((yard_idx != ref_idx).sum())
    print(f"{name} bf16 vs fp32 reference: mismatches HIP {raw}/{n} | yardstick {yraw}/{n}; mean|bounded err| HIP {float(err.mean()):.4f} | "
          f"yardstick {float(yerr.mean()):.4f}; max HIP {float(err.max()):.4f} | yardstick {float(yerr.max()):.4f}")
    assert raw <= yraw + bf16_slack(n, yraw) + extra
    for tau in TAU_LIST:
        safe = margin > tau
        mine, yard = int((idx[safe] != ref_idx[safe]).sum()), int((yard_idx[safe] != ref_idx[safe]).sum())
        print(f"   margin > {tau}: {int(safe.sum())} tokens, mismatches HIP {mine} | yardstick {yard}")
        assert mine <= yard + bf16_slack(int(safe.sum()), yard) + extra
    assert float(err.mean()) <= r_mean * float(yerr.mean())
    assert float(err.max()) <= r_max * float(yerr.max())
    # exact wherever the fp32 value is further from a rounding boundary than the YARDSTICK's maximum error
    safe = margin > float(yerr.max())
    assert np.array_equal(idx[safe], ref_idx[safe])


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
    assert_bf16_not_worse_than_yardstick(name, idx, bounded, d["indices"], torch.from_numpy(d["bounded"]), d["indices_refbf16"],
                                         torch.from_numpy(d["bounded_refbf16"]))
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


def test_rotary_factors_by_position_id_equal_the_table_path(monkeypatch):
    """ttv_batch.rope_ids / rope_base (8 bytes per row + the L2-resident base table, gathered inside the width-256 to_qkv kernel)
    against the [L,64] fp32 table path (TTV_ROPE_IDS=0): the same cos / sin values, so tokens and pixels must be bit-equal."""
    from titok_video_amd import plan as P
    model = build(torch.bfloat16)
    shapes = [(8, 32, 48), (4, 16, 16), (16, 64, 32), (16, 128, 128)]
    counts = [5, 1, 17, 128]
    clips = synthetic_clips(shapes, seed=23, dtype=torch.bfloat16, device=DEV)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("TTV_ROPE_IDS", flag)
        P._plan_cache.clear()
        plan = P.get_plan(shapes, counts, (4, 8, 8), torch.device(DEV))
        assert bool(plan.batch_for(4, 2).rope_ids) == (flag == "1")
        with torch.no_grad():
            recon, out = model(clips, counts)
        outs.append(([r.clone() for r in recon], out["indices"].clone()))
    P._plan_cache.clear()
    assert torch.equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][0], outs[1][0]):
        assert torch.equal(a, b)


def test_to_qkv_kernels_agree_inside_the_towers():
    """With the folded pre-norm (the towers' path) k_qkv256 / k_qkv256ws take the row statistic from v_dot2c_f32_bf16 partial sums and fold
    it into the rotary factors, k_gemm_k256 from an fma chain and scales the sums: the same values up to fp32 rounding of rstd, i.e. an
    occasional last-place flip of a bf16 q / k / v element, which four bf16 layers then amplify (two bf16 executions of this model differ
    by ~0.008 mean / ~0.1 max in the pre-rounding latents: tools/qkv256_tower_stats.py).  So the gate is the fp32 oracle: the forward
    through the new kernels is not less accurate than the forward through k_gemm_k256 (mean 1.1 x, max 1.25 x), the two stay close to
    each other on average, and the two new kernels - which share their arithmetic - give identical forwards."""
    model = build(torch.bfloat16)
    shapes = [(8, 32, 48), (4, 16, 16), (16, 64, 32), (16, 128, 128)]
    counts = [5, 1, 17, 128]
    clips = synthetic_clips(shapes, seed=29, dtype=torch.bfloat16, device=DEV)
    from titok_video_amd import _lib
    lib = _lib.lib()
    outs = []
    try:
        for bits in (0, 1 << 15, 1 << 17):
            lib.ttv_debug_set(bits)
            with torch.no_grad():
                recon, out = model(clips, counts)
                model.encode(clips, counts, want_bounded=True)
            torch.cuda.synchronize()
            outs.append(([r.float().clone() for r in recon], out["indices"].clone(), model.last_bounded.float().cpu().clone()))
    finally:
        lib.ttv_debug_set(0)
    with torch.no_grad():
        _r, _i, _z, ref_b = O.titok_forward([c.float().cpu() for c in clips], counts, seeded_titok_state(0), LEVELS)
    err = [(o[2] - ref_b).abs() for o in outs]
    print(f"bounded-latent error against the fp32 oracle: k_qkv256 mean {err[0].mean():.5f} max {err[0].max():.4f}; "
          f"k_gemm_k256 mean {err[1].mean():.5f} max {err[1].max():.4f}")
    assert float(err[0].mean()) <= 1.1 * float(err[1].mean()), (float(err[0].mean()), float(err[1].mean()))
    assert float(err[0].max()) <= 1.25 * float(err[1].max()), (float(err[0].max()), float(err[1].max()))
    assert float((outs[0][2] - outs[1][2]).abs().mean()) < 0.02
    assert torch.equal(outs[0][1], outs[2][1]) and all(torch.equal(x, y) for x, y in zip(outs[0][0], outs[2][0])), \
        "k_qkv256 and k_qkv256ws share their arithmetic: identical forward expected"


@pytest.mark.parametrize("mode", ["bf16", "fp32", "split3"])
def test_encoder_last_layer_on_latent_rows_only_changes_no_bit(mode):
    """The encoder's output is read from its latent rows (blocks.py:101-103).  With `ttv_batch.qblocks_latent` the last layer runs its
    attention for the latent query rows only and out_proj / the KEEL norms / the feed-forward on the gathered latent rows; every kernel
    behind the attention is row-wise and the attention computes a query block by itself, so indices, codes and the pre-rounding latents
    are the bits of the all-rows forward (ttv_debug_set bit 19 switches the shortcut off).  Mixed shapes: token counts that do and do
    not fill a 128-row query block, a clip with more tokens than one block."""
    from titok_video_amd import _lib
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    model = build(dt)
    if mode == "split3":
        model.set_index_exact("split3")
    shapes = [(8, 32, 48), (4, 16, 16), (16, 64, 32), (16, 128, 128), (16, 128, 128)]
    counts = [5, 1, 17, 128, 200]
    clips = synthetic_clips(shapes, seed=31, dtype=dt, device=DEV)
    lib = _lib.lib()
    outs = []
    try:
        for bits in (0, 1 << 19):
            lib.ttv_debug_set(bits)
            with torch.no_grad():
                codes, info = model.encode(clips, counts, want_bounded=True)
            torch.cuda.synchronize()
            outs.append((codes.clone(), info["indices"].clone(), model.last_bounded.clone()))
    finally:
        lib.ttv_debug_set(0)
    assert torch.equal(outs[0][1], outs[1][1]), "indices differ"
    assert torch.equal(outs[0][2], outs[1][2]), "pre-rounding latents differ"
    assert torch.equal(outs[0][0], outs[1][0]), "codes differ"


def test_decoder_last_layer_without_latent_only_query_blocks_changes_no_bit(monkeypatch):
    """The decoder's output is read from its patch rows (blocks.py:171).  With `ttv_batch.qblocks_patch` the last layer's attention
    skips the 128-row query blocks that hold latent rows only; everything behind the attention is row-wise and the tail gathers patch
    rows, so the reconstructions are the bits of the all-blocks forward (ttv_debug_set bit 21 switches the shortcut off).  Token counts
    of one block exactly (128: block 0 skipped), more than one block (200: block 0 skipped, block 1 mixed), and less than one (17:
    nothing to skip)."""
    from titok_video_amd import _lib
    from titok_video_amd.plan import BatchPlan
    monkeypatch.setenv("TTV_ATTN_SPLIT", "0")        # full items at this small batch too (the shortcut is for tables of full items)
    model = build(torch.bfloat16)
    shapes = [(16, 128, 128), (16, 128, 128), (16, 64, 64), (8, 32, 48)]
    counts = [128, 200, 17, 128]
    plan = BatchPlan(shapes, counts, (4, 8, 8), torch.device(DEV))
    full, part = plan.attention_table(4, 2), plan.attention_table_patch(4, 2)
    n = lambda t: int((t[:, 0] >= 0).sum())
    assert part is not None and n(part) == n(full) - 4 * (1 + 1 + 0 + 1)      # one latent-only block per clip with K >= 128, 4 q-heads
    clips = synthetic_clips(shapes, seed=33, dtype=torch.bfloat16, device=DEV)
    lib = _lib.lib()
    with torch.no_grad():
        codes, info = model.encode(clips, counts)
    outs = []
    try:
        for bits in (0, 1 << 21):
            lib.ttv_debug_set(bits)
            with torch.no_grad():
                recon = model.decode(codes, counts, shapes)
            torch.cuda.synchronize()
            outs.append([r.clone() for r in recon])
    finally:
        lib.ttv_debug_set(0)
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), "reconstructions differ"
    assert all(torch.isfinite(a.float()).all() for a in outs[0])


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
    # the whole 4-layer stack (layer 0 pre-LN residual, layers >= 1 KEEL: x = post_ln(alpha * x + f(x)), alpha = 2 * layers,
    # transformer.py:126-146), chained from the single ops, against the reference's `model_layers(x)`
    alpha = 8.0
    xs = x.clone()
    y32 = torch.empty(Lr, dm, device=DEV)
    tbl = plan.attention_table(4, 2)

    def wt(name):
        return w[p + name].data_ptr()
    for i in range(4):
        _lib.check(lib.ttv_rmsnorm(xs.data_ptr(), 1, dm, None, xn.data_ptr(), 1, dm, None, wt(f"attn_layer.{i}.pre_ln.weight"), Lr, dm, 1e-5, S), "n")
        _lib.check(lib.ttv_linear_qkv_rope(xn.data_ptr(), dm, wt(f"attn_layer.{i}.to_qkv.weight"), dm, qkv.data_ptr(), 2 * dm + 2 * gq, Lr, dm, gq, plan.rope_cs.data_ptr(), 1, S), "q")
        _lib.check(lib.ttv_attention(qkv.data_ptr(), 2 * dm + 2 * gq, ao.data_ptr(), dm, plan.cu_dev.data_ptr(), tbl.data_ptr(), tbl.shape[0], 4, 2, 64, 1, 1, S), "a")
        if i == 0:
            _lib.check(lib.ttv_linear_residual(ao.data_ptr(), dm, wt("attn_layer.0.out_proj.weight"), dm, xs.data_ptr(), dm, 1.0, xs.data_ptr(), dm, 0, Lr, dm, dm, 1, S), "o")
        else:
            _lib.check(lib.ttv_linear_residual(ao.data_ptr(), dm, wt(f"attn_layer.{i}.out_proj.weight"), dm, xs.data_ptr(), dm, alpha, y32.data_ptr(), dm, 1, Lr, dm, dm, 1, S), "o")
            _lib.check(lib.ttv_rmsnorm(y32.data_ptr(), 1, dm, None, xs.data_ptr(), 1, dm, None, wt(f"attn_post_ln.{i - 1}.weight"), Lr, dm, 1e-5, S), "n")
        _lib.check(lib.ttv_rmsnorm(xs.data_ptr(), 1, dm, None, xn.data_ptr(), 1, dm, None, wt(f"ffd_layer.{i}.norm.weight"), Lr, dm, 1e-5, S), "n")
        _lib.check(lib.ttv_linear_geglu(xn.data_ptr(), dm, wt(f"ffd_layer.{i}.w12.weight"), dm, h.data_ptr(), I, Lr, I, dm, 1, S), "g")
        if i == 0:
            _lib.check(lib.ttv_linear_residual(h.data_ptr(), I, wt("ffd_layer.0.w3.weight"), I, xs.data_ptr(), dm, 1.0, xs.data_ptr(), dm, 0, Lr, dm, I, 1, S), "o")
        else:
            _lib.check(lib.ttv_linear_residual(h.data_ptr(), I, wt(f"ffd_layer.{i}.w3.weight"), I, xs.data_ptr(), dm, alpha, y32.data_ptr(), dm, 1, Lr, dm, I, 1, S), "o")
            _lib.check(lib.ttv_rmsnorm(y32.data_ptr(), 1, dm, None, xs.data_ptr(), 1, dm, None, wt(f"ffd_post_ln.{i - 1}.weight"), Lr, dm, 1e-5, S), "n")
    np.testing.assert_allclose(xs.cpu().numpy(), d["stack"], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("size", ["small", "base", "large"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_other_model_sizes_match_oracle(size, dtype):
    """get_model_dims sizes beyond tiny (utils.py:8-23): widths 512/768/1024, GQA 8:2 / 12:4 / 16:4, GEGLU inner 1376 /
    2048 / 2752 (1376 is not a multiple of the 64-wide feature tiles) - generic (unfused) kernel path vs the CPU oracle."""
    sd = seeded_titok_state(3, encoder_size=size, decoder_size=size, gain=3.0)
    m = TiTok(config(enc=size, dec=size))
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV, dtype).eval()
    shapes, counts = [(4, 16, 16), (8, 16, 24), (4, 32, 16)], [128, 128, 128]     # >= 128 tokens per clip: 384 in all
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
        # yardstick: the REFERENCE's own modules run in bf16 on these inputs (tests/golden/titok_sizes.npz, make_golden_sizes.py).
        # Round 2 used the CPU oracle evaluated in bf16 here; that is a more accurate bf16 execution than the reference's (mean error
        # 0.0098 / 0.0170 against the reference's own 0.0143 / 0.0246 for small / base), i.e. a bar the reference does not meet.
        g = np.load(os.path.join(G, "titok_sizes.npz"))
        assert g["shapes"].tolist() == [list(s_) for s_ in shapes] and g["counts"].tolist() == counts and int(g["clip_seed"]) == 13
        assert np.array_equal(ref_idx.numpy(), g[f"{size}_indices"])                 # the oracle's fp32 run IS the reference's
        assert_bf16_not_worse_than_yardstick(size, dd["indices"].cpu().numpy(), m.last_bounded.cpu(), g[f"{size}_indices"],
                                             torch.from_numpy(g[f"{size}_bounded"]), g[f"{size}_indices_refbf16"],
                                             torch.from_numpy(g[f"{size}_bounded_refbf16"]))
        assert perr < 0.08 * max(1.0, scale)


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
    # the extremes of the loader's ranges (largest grid with K = 128 and with K = 1) plus two full-K clips: 386 tokens
    shapes, counts = [(16, 168, 168), (8, 128, 128), (16, 168, 168), (4, 16, 16), (8, 16, 24)], [128, 1, 1, 128, 128]
    clips = synthetic_clips(shapes, seed=77, dtype=torch.float32, device=DEV)
    model = build(torch.float32)
    with torch.no_grad():
        codes, od = model.encode(clips, counts, want_bounded=True)
        recon = model.decode(codes, counts, shapes)
    sd = seeded_titok_state(0)
    ref_recon, ref_idx, _ref_z, ref_bounded = O.titok_forward([c.cpu() for c in clips], counts, sd, LEVELS)
    idx = od["indices"].cpu()
    assert idx.shape == (386,)
    safe = O.fsq_margin(ref_bounded) > TAU_F32
    assert int(safe.sum()) >= 360
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
    number of key tiles): not worse than the REFERENCE's own bf16 run on the same inputs (tests/golden/titok_extremes.npz,
    make_golden_sizes.py; fixed thresholds, see the header)."""
    # the extremes of the loader's ranges (largest grid with K = 128 and with K = 1) plus two full-K clips: 386 tokens
    shapes, counts = [(16, 168, 168), (8, 128, 128), (16, 168, 168), (4, 16, 16), (8, 16, 24)], [128, 1, 1, 128, 128]
    clips32 = synthetic_clips(shapes, seed=77, dtype=torch.float32, device="cpu")
    model = build(torch.bfloat16)
    clips = [c.to(DEV, torch.bfloat16) for c in clips32]
    with torch.no_grad():
        codes, od = model.encode(clips, counts, want_bounded=True)
        recon = model.decode(codes, counts, shapes)
    sd = seeded_titok_state(0)
    g = np.load(os.path.join(G, "titok_extremes.npz"))
    assert g["shapes"].tolist() == [list(s_) for s_ in shapes] and g["counts"].tolist() == counts and int(g["clip_seed"]) == 77
    assert_bf16_not_worse_than_yardstick("sampling corners", od["indices"].cpu().numpy(), model.last_bounded.cpu(), g["indices"],
                                         torch.from_numpy(g["bounded"]), g["indices_refbf16"], torch.from_numpy(g["bounded_refbf16"]))
    dec_ref = O.titok_decode_indices(od["indices"].cpu(), shapes, counts, sd, LEVELS)
    for r, ref in zip(recon, dec_ref):
        assert float((r.float().cpu() - ref).abs().max()) < PIX_TOL_BF16 * 1.5


def test_batch32_bf16_equals_eight_batches_of_four_and_the_fixture(monkeypatch):
    """BASELINE config #2 at its full size (32 x 16x128x128, K = 128: L = 36 864 rows, the tile shapes the benchmark selects).
    With one attention work-table regime for both sizes (full items only) the batch equals eight batch-4 calls bit for bit - a
    clip's result does not depend on its batch mates - and its first four clips are the fixture's: same parity statistics as
    the batch-4 fixture test."""
    monkeypatch.setenv("TTV_ATTN_SPLIT", "0")
    d = np.load(os.path.join(G, "titok_cfg1.npz"))
    model = build(torch.bfloat16)
    shapes, counts = [(16, 128, 128)] * 32, [128] * 32
    clips = synthetic_clips(shapes, seed=int(d["clip_seed"]), dtype=torch.bfloat16, device=DEV)
    with torch.no_grad():
        codes, od = model.encode(clips, counts, want_bounded=True)
        bounded = model.last_bounded.clone()
        recon = model.decode(codes, counts, shapes)
        for j in range(8):
            c4, o4 = model.encode(clips[4 * j:4 * j + 4], counts[:4], want_bounded=True)
            assert torch.equal(o4["indices"], od["indices"][512 * j:512 * (j + 1)])
            assert torch.equal(model.last_bounded, bounded[512 * j:512 * (j + 1)])
            r4 = model.decode(c4, counts[:4], shapes[:4])
            for a, b in zip(r4, recon[4 * j:4 * j + 4]):
                assert torch.equal(a, b)
    assert_bf16_not_worse_than_yardstick("cfg1 at batch 32", od["indices"][:512].cpu().numpy(), bounded[:512].cpu(), d["indices"],
                                         torch.from_numpy(d["bounded"]), d["indices_refbf16"], torch.from_numpy(d["bounded_refbf16"]))


def test_batch32_fp32_indices_equal_the_reference():
    """The float32 towers (exact-fp32 MFMA kernels) at the benchmark batch: every token of the fixture's four clips whose rounding
    margin exceeds 1e-3 has the reference's index (the same check bench.py reports as parity.fp32_exact)."""
    d = np.load(os.path.join(G, "titok_cfg1.npz"))
    model = build(torch.float32)
    shapes, counts = [(16, 128, 128)] * 32, [128] * 32
    clips = synthetic_clips(shapes, seed=int(d["clip_seed"]), dtype=torch.float32, device=DEV)
    with torch.no_grad():
        codes, od = model.encode(clips, counts, want_bounded=True)
    idx = od["indices"][:512].cpu().numpy()
    margin = O.fsq_margin(torch.from_numpy(d["bounded"])).numpy()
    safe = margin > TAU_F32
    assert np.array_equal(idx[safe], d["indices"][safe])
    assert (idx != d["indices"]).sum() <= 2, (idx != d["indices"]).sum()
    np.testing.assert_allclose(model.last_bounded[:512].cpu().numpy(), d["bounded"], rtol=0, atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_token_and_patch_initialisation_rows(dtype):
    """Row a6 of the scope table: the input of layer 0 (blocks.py:95-97 encoder, :165-167 decoder) from the two row kernels alone -
    constant rows RMSNorm(mask_token * 1) * gain and the decoder's latent rows RMSNorm(proj_in(codes) + mask_token) * gain -
    against the same expressions evaluated in float64 on the host."""
    from titok_video_amd import _lib
    lib, S = _lib.lib(), _lib.stream_ptr(torch.device(DEV))
    code = _lib.dtype_code(dtype)
    sd = seeded_titok_state(0)
    dm, C, rows = 256, 5, 37
    gen = torch.Generator().manual_seed(9)
    rows_map = torch.randperm(64, generator=gen)[:rows].to(torch.int32)
    x = torch.zeros(64, dm, dtype=dtype, device=DEV)
    mt = sd["encoder.mask_token"].reshape(1).to(DEV)
    g_t = sd["encoder.ln_pre_t.weight"].to(DEV)
    rm_dev = rows_map.to(DEV)
    _lib.check(lib.ttv_fill_const_rows(x.data_ptr(), code, dm, rm_dev.data_ptr(), rows, dm, mt.data_ptr(), g_t.data_ptr(), 1e-5, S), "fill")
    m = mt.to(dtype).double().cpu()
    ref = (m / torch.sqrt(m * m + 1e-5)) * g_t.double().cpu()
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    got = x.double().cpu()
    assert float((got[rows_map.long()] - ref[None, :]).abs().max()) < tol
    untouched = torch.ones(64, dtype=torch.bool)
    untouched[rows_map.long()] = False
    assert float(got[untouched].abs().max()) == 0.0
    # decoder latent rows
    codes = O.fsq_indices_to_codes(torch.randint(0, 4375, (rows,), generator=gen, dtype=torch.int32), LEVELS).to(dtype)
    wv, bv = sd["decoder.proj_in.weight"].to(dtype), sd["decoder.proj_in.bias"].to(dtype)
    mtd, gd = sd["decoder.mask_token"].reshape(1).to(DEV), sd["decoder.ln_pre_t.weight"].to(DEV)
    x.zero_()
    codes_d, wv_d, bv_d = codes.to(DEV), wv.to(DEV), bv.to(DEV)     # kept alive: raw pointers go to the library
    _lib.check(lib.ttv_decoder_embed(codes_d.data_ptr(), C, wv_d.data_ptr(), bv_d.data_ptr(), mtd.data_ptr(), gd.data_ptr(), x.data_ptr(),
                                     code, dm, rm_dev.data_ptr(), rows, dm, 1e-5, S), "embed")
    torch.cuda.synchronize()
    hh = codes.double() @ wv.double().t() + bv.double() + mtd.to(dtype).double().cpu()
    ref = hh * torch.rsqrt(hh.pow(2).mean(-1, keepdim=True) + 1e-5) * gd.double().cpu()
    got = x.double().cpu()[rows_map.long()]
    assert float((got - ref).abs().max()) < (2e-5 if dtype == torch.float32 else 6e-2), float((got - ref).abs().max())


def test_forward_pipeline_cold_start_unseen_shapes():
    """A FRESH model (no packed weights yet) and batch shapes no plan exists for, submitted straight into a depth-2 pipeline:
    the weight pack and the plans are built on the first stream while the second starts - it must wait for them (events recorded by
    the pack / plan) - and the callers' input tensors are dropped right after submit (record_stream keeps them alive)."""
    from titok_video_amd.pipeline import ForwardPipeline
    specs = [([(4, 24, 40), (8, 16, 16)], [4, 6]), ([(8, 40, 24)], [11]), ([(4, 8, 40), (4, 24, 8), (4, 16, 48)], [2, 2, 5]), ([(8, 24, 24)], [7]),
             ([(4, 24, 40), (8, 16, 16)], [4, 6]), ([(12, 16, 8)], [3])]
    cold = build(torch.bfloat16, seed=4)
    pipe = ForwardPipeline(cold, depth=2)
    tickets = []
    for i, (shapes, counts) in enumerate(specs):
        clips = synthetic_clips(shapes, seed=300 + i, dtype=torch.bfloat16, device=DEV)
        tickets.append(pipe.submit(clips, counts))
        del clips                                              # the side stream may not have started reading them yet
    outs = [pipe.result(t) for t in tickets]
    torch.cuda.synchronize()
    warm = build(torch.bfloat16, seed=4)
    for i, ((shapes, counts), (recon, info)) in enumerate(zip(specs, outs)):
        clips = synthetic_clips(shapes, seed=300 + i, dtype=torch.bfloat16, device=DEV)
        with torch.no_grad():
            r0, o0 = warm(clips, counts)
        assert torch.equal(o0["indices"], info["indices"])
        for a, b in zip(r0, recon):
            assert torch.equal(a, b)


def test_weight_updates_through_data_reach_the_kernels():
    """`.data` writes do not bump a parameter's version counter, so the packed-weight cache cannot see them: init_weights (which
    writes through .data) and load_state_dict invalidate the packs themselves, other `.data` updates call invalidate_packs()."""
    from titok_video_amd.model.base.utils import init_weights
    model = build(torch.bfloat16)
    shapes, counts = [(4, 16, 16), (8, 16, 24)], [3, 6]
    clips = synthetic_clips(shapes, seed=8, dtype=torch.bfloat16, device=DEV)
    with torch.no_grad():
        r0, _ = model(clips, counts)
        model.decoder.proj_out.weight.data.mul_(2.0)           # invisible to the version counters
        model.decoder.invalidate_packs()
        r1, _ = model(clips, counts)
        assert not torch.equal(r0[0], r1[0])
        model.decoder.proj_out.weight.mul_(0.5)                # in-place under no_grad: bumps the version, repacked by itself
        r2, _ = model(clips, counts)
        assert torch.equal(r0[0], r2[0])
        model.apply(init_weights)                              # trunc_normal_(weight.data): must not leave the old packs in use
        r3, o3 = model(clips, counts)
        assert not torch.equal(r0[0], r3[0])
        fresh = TiTok(config()).to(DEV, torch.bfloat16).eval()
        fresh.load_state_dict(model.state_dict(), strict=True)
        r4, o4 = fresh(clips, counts)
        assert torch.equal(o3["indices"], o4["indices"]) and torch.equal(r3[0], r4[0])


# ---------------------------------------------------------------------------------------------- BASELINE config #4
def _base_setup(dtype):
    d = np.load(os.path.join(G, "titok_base_cfg4.npz"))
    levels = d["levels"].tolist()
    m = TiTok(config(levels=levels, enc="base", dec="base"))
    m.load_state_dict(seeded_titok_state(int(d["weight_seed"]), "base", "base", gain=float(d["weight_gain"])), strict=True)
    m = m.to(DEV, dtype).eval()
    shape, count = tuple(d["shape"].tolist()), int(d["count"])
    clips = synthetic_clips([shape], seed=int(d["clip_seed"]), dtype=dtype, device=DEV)
    return d, levels, m, shape, count, clips


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "titok_base_cfg4.npz")), reason="fixture not generated")
def test_base_cfg4_fp32_vs_reference():
    """BASELINE config #4 at its real size: base towers (d = 768, 12 layers, heads 12/4), ONE 32x256x256 clip, K = 1024 latent tokens -
    a 9216-row sequence (144 key tiles, 72 query blocks per head), FSQ [8,8,8,6,5] - against the reference's own fp32 run
    (tests/golden/make_golden_base.py).  float32 towers: the reference's index on every token away from a rounding boundary."""
    d, levels, m, shape, count, clips = _base_setup(torch.float32)
    with torch.no_grad():
        codes, od = m.encode(clips, [count], want_bounded=True)
        recon = m.decode(codes, [count], [shape])
    idx = od["indices"].cpu().numpy()
    margin = O.fsq_margin(torch.from_numpy(d["bounded"])).numpy()
    safe = margin > TAU_F32
    print(f"base cfg4 fp32: {int((idx != d['indices']).sum())} of {idx.size} indices differ; {int(safe.sum())} tokens with margin > {TAU_F32}; "
          f"max |bounded err| {float(np.abs(m.last_bounded.cpu().numpy() - d['bounded']).max()):.2e}")
    assert np.array_equal(idx[safe], d["indices"][safe])
    assert (idx != d["indices"]).sum() <= 3
    np.testing.assert_allclose(m.last_bounded.cpu().numpy(), d["bounded"], rtol=0, atol=5e-3)
    np.testing.assert_allclose(recon[0].cpu().numpy()[:, ::4, ::8, ::8], d["recon_sample"], rtol=0, atol=2e-2)
    assert abs(float(recon[0].double().std()) - float(d["recon_std"])) < 2e-3


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "titok_base_cfg4.npz")), reason="fixture not generated")
def test_base_cfg4_bf16_vs_reference():
    """The same clip through the bf16 kernels (general-K GEMMs at widths 768 / 2048, the attention kernel at 12/4 heads and 144 key
    tiles): not worse than the reference's own bf16 run (fixture keys *_refbf16), fixed thresholds as for the tiny fixtures."""
    d, levels, m, shape, count, clips = _base_setup(torch.bfloat16)
    with torch.no_grad():
        codes, od = m.encode(clips, [count], want_bounded=True)
    assert_bf16_not_worse_than_yardstick("base cfg4", od["indices"].cpu().numpy(), m.last_bounded.cpu(), d["indices"], torch.from_numpy(d["bounded"]),
                                         d["indices_refbf16"], torch.from_numpy(d["bounded_refbf16"]))
    ref_codes = O.fsq_indices_to_codes(torch.from_numpy(d["indices"]), levels).to(torch.bfloat16).to(DEV)
    with torch.no_grad():
        rec = m.decode(ref_codes, [count], [shape])
    mine = rec[0].float().cpu()[:, ::4, ::8, ::8].flatten()
    ref, ref16 = torch.from_numpy(d["recon_sample"]).flatten(), torch.from_numpy(d["recon_sample_refbf16"]).flatten()
    e_mine, e_ref16 = float((mine - ref).norm() / ref.norm()), float((ref16 - ref).norm() / ref.norm())
    print(f"   decoder rel. error vs fp32 reference: HIP {e_mine:.5f} | reference-in-bf16 {e_ref16:.5f}")
    assert e_mine <= 1.15 * e_ref16 + 1e-3


def test_base_cfg5_mx_fp8_towers_with_the_16384x64_codebook_on_one_full_size_clip():
    """BASELINE config #5 as ONE thing (missing #1): base towers, one 32x256x256 clip, K = 1024 latent tokens, the L2
    quantiser with a 16384 x 64 codebook wired into TiTok, all four linears of every layer in block-scaled (MX) e4m3 - against the CPU
    oracle towers (fp32) + the cdist oracle.  Not reference-pinned by construction (the reference has neither an L2 quantiser nor fp8):
    the oracle restates the model, the MX arithmetic is pinned kernel by kernel in tests/test_hip_fp8.py.
    Stated tolerance (e4m3 keeps 3 mantissa bits, 96 quantised linears in a row): pre-quantisation tokens z within 0.20 relative
    (Frobenius) of the fp32 oracle (measured 0.111; the bf16 towers 0.019), decoder reconstruction of the ORACLE's codes within 0.15
    relative (measured 0.118; bf16 0.020); at least 60 % of the token indices equal to the fp32 oracle's (measured 81.9 %; bf16 towers
    96.8 %; MX against bf16 towers 82.6 %): a 16384-entry L2 codebook is far more forgiving than FSQ's rounding boundaries."""
    from types import SimpleNamespace
    from oracle import vq_oracle as V
    n_entries, width, count, shape = 16384, 64, 1024, (32, 256, 256)
    cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(
        patch_size=[4, 8, 8], fsq_levels=None, quantizer="l2", codebook_size=n_entries, token_size=width, encoder_size="base", decoder_size="base")))
    sd = seeded_titok_state(0, "base", "base", token_size=width, gain=3.0)
    cb = torch.randn(n_entries, width, generator=torch.Generator().manual_seed(5)) * 1.5
    clip_cpu = synthetic_clips([shape], seed=4044)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        z_ref = O.encoder_forward(clip_cpu, [count], sd, "base", prefix="encoder.")
        idx_ref, _, _gap = V.l2_argmin(z_ref, cb)
        rec_ref = O.decoder_forward(cb[idx_ref.long()], [count], [shape], sd, "base", prefix="decoder.")[0]
    res = {}
    for mode in (False, "mx"):
        m = TiTok(cfg)
        m.load_state_dict({**sd, "quantize.codebook": cb}, strict=True)
        m = m.to(DEV, torch.bfloat16).eval()
        m.encoder.fp8_linears = m.decoder.fp8_linears = mode
        clips = [c.to(DEV, torch.bfloat16) for c in clip_cpu]
        with torch.no_grad():
            z = m.encoder.run(clips, [count], None, None, want_z=True)["z"].float().cpu()
            recon, info = m(clips, [count])                                           # the whole forward, as bench.py --config base5 times it
            rec_o = m.decode(cb[idx_ref.long()].to(DEV, torch.bfloat16), [count], [shape])[0].float().cpu()
        assert recon[0].shape == (3,) + shape and info["indices"].shape == (count,)
        res[mode] = dict(z=float((z - z_ref).norm() / z_ref.norm()), idx=info["indices"].cpu(),
                         dec=float((rec_o - rec_ref).norm() / rec_ref.norm()))
    agree = lambda a, b: float((a == b).float().mean())
    print(f"base cfg5, one 32x256x256 clip: z rel. error bf16 {res[False]['z']:.4f} | mx-fp8 {res['mx']['z']:.4f}; decoder rel. error on the oracle's codes "
          f"bf16 {res[False]['dec']:.4f} | mx-fp8 {res['mx']['dec']:.4f}; indices equal to the fp32 oracle's: bf16 {agree(res[False]['idx'], idx_ref):.3f} | "
          f"mx-fp8 {agree(res['mx']['idx'], idx_ref):.3f}; mx-fp8 vs bf16 towers {agree(res['mx']['idx'], res[False]['idx']):.3f}")
    assert res["mx"]["z"] < 0.20 and res["mx"]["dec"] < 0.15
    assert res[False]["z"] < 0.05
    assert agree(res["mx"]["idx"], idx_ref) > 0.60 and agree(res[False]["idx"], idx_ref) > 0.90
