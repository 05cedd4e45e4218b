"""Pin the CPU oracle (oracle/titok_oracle.py) against fixtures generated from the reference's own code
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import titok_oracle as O
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_fsq_kat(tag):
    d = load("fsq_kat.npz")
    levels = d[f"levels_{tag}"].tolist()
    z = torch.from_numpy(d[f"z_{tag}"])
    codes, idx, bounded = O.fsq_forward(z, levels)
    assert np.array_equal(idx.numpy(), d[f"indices_{tag}"])           # bit-exact integer indices
    assert np.array_equal(codes.numpy(), d[f"codes_{tag}"])
    assert np.array_equal(bounded.numpy(), d[f"bounded_{tag}"])
    n = int(np.prod(levels))
    cb = O.fsq_indices_to_codes(torch.arange(n, dtype=torch.int32), levels)
    assert np.array_equal(cb.numpy(), d[f"codebook_{tag}"])
    # round trip + range
    _, idx2, _ = O.fsq_forward(torch.atanh(cb.clamp(-0.999, 0.999)) * 0 + cb * 10, levels)
    assert idx.min() >= 0 and idx.max() < n


def test_fsq_is_lattice_argmin_away_from_ties():
    levels = [7, 5, 5, 5, 5]
    g = torch.Generator().manual_seed(3)
    z = torch.randn(2000, 5, generator=g) * 1.5
    codes, idx, bounded = O.fsq_forward(z, levels)
    n = int(np.prod(levels))
    lattice = O.fsq_indices_to_codes(torch.arange(n, dtype=torch.int32), levels) * (torch.tensor(levels) // 2)
    arg = torch.cdist(bounded, lattice.float()).argmin(-1).to(torch.int32)
    keep = O.fsq_margin(bounded) > 1e-4
    assert torch.equal(arg[keep], idx[keep])


def test_rope_table_and_rotary():
    d = load("rope_kat.npz")
    for i in range(4):
        cos, sin = O.rope_table(d[f"grids_{i}"].tolist(), d[f"counts_{i}"].tolist())
        assert np.array_equal(cos.numpy(), d[f"cos_{i}"])
        assert np.array_equal(sin.numpy(), d[f"sin_{i}"])
    cos, sin = O.rope_table(d["grids_1"].tolist(), d["counts_1"].tolist())
    out = O.apply_rotary(torch.from_numpy(d["rot_q"]), cos, sin)
    np.testing.assert_allclose(out.numpy(), d["rot_out"], rtol=0, atol=1e-6)
    # RoPE leaves the last 4 dims of each head untouched (rope.py:24,40)
    assert np.array_equal(out.numpy()[..., 60:], d["rot_q"][..., 60:])


def test_patchify_roundtrip():
    d = load("patch_kat.npz")
    clip = torch.from_numpy(d["clip"])
    p = O.patchify(clip, (4, 8, 8))
    assert np.array_equal(p.numpy(), d["patches"])
    assert torch.equal(O.unpatchify(p, (2, 2, 3), (4, 8, 8), 3), clip)


def test_codebook_scores():
    d = load("codebook_kat.npz")
    sizes = d["sizes"].tolist()
    samples = list(torch.split(torch.from_numpy(d["flat"]), sizes))
    n = int(d["codebook_size"])
    usage, ent, _ = O.codebook_scores(samples[-n:], n)     # FIFO keeps the last `codebook_size` samples
    assert abs(usage - float(d["usage"])) < 1e-4
    assert abs(ent - float(d["entropy"])) < 1e-5


def test_blocks_kat():
    d = load("blocks_kat.npz")
    sd = seeded_titok_state(int(d["weight_seed"]))
    grids, counts = d["grids"].tolist(), d["counts"].tolist()
    cu = [0]
    for g, k in zip(grids, counts):
        cu.append(cu[-1] + int(np.prod(g)) + k)
    cos, sin = O.rope_table(grids, counts)
    x = torch.from_numpy(d["x"])
    a1 = O.attn_sublayer(x, sd, "encoder.model_layers.attn_layer.1.", (4, 2), cos, sin, cu)
    f1 = O.geglu_sublayer(x, sd, "encoder.model_layers.ffd_layer.1.")
    st = O.transformer_stack(x, sd, "encoder.model_layers.", 4, (4, 2), cos, sin, cu)
    np.testing.assert_allclose(a1.numpy(), d["attn1"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(f1.numpy(), d["ffd1"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(st.numpy(), d["stack"], rtol=1e-4, atol=1e-4)


def _run_small(name):
    d = load(name)
    sd = seeded_titok_state(int(d["weight_seed"]))
    if "shapes" in d:
        shapes, counts = d["shapes"].tolist(), d["counts"].tolist()
    else:
        shapes, counts = [d["shape"].tolist()], [int(d["count"])]
    clips = synthetic_clips(shapes, seed=int(d["clip_seed"]))
    with torch.no_grad():
        recon, idx, z, bounded = O.titok_forward(clips, counts, sd, [7, 5, 5, 5, 5])
    return d, recon, idx, z, bounded


def test_titok_small_matches_reference():
    d, recon, idx, z, bounded = _run_small("titok_small.npz")
    assert np.array_equal(idx.numpy(), d["indices"])                       # token indices bit-exact
    np.testing.assert_allclose(z.numpy(), d["z"], rtol=1e-4, atol=1e-4)
    for i, r in enumerate(recon):
        np.testing.assert_allclose(r.numpy(), d[f"recon_{i}"], rtol=1e-3, atol=1e-3)


def test_packing_invariance_and_single():
    d, recon, idx, z, bounded = _run_small("titok_single.npz")
    assert np.array_equal(idx.numpy(), d["indices"])
    np.testing.assert_allclose(recon[0].numpy(), d["recon"], rtol=1e-3, atol=1e-3)


def test_titok_cfg1_matches_reference():
    """BASELINE config #1: 4 x 16x128x128 clips, K=128, tiny, fp32 CPU."""
    d, recon, idx, z, bounded = _run_small("titok_cfg1.npz")
    ref_idx = d["indices"]
    margin = O.fsq_margin(torch.from_numpy(d["bounded"]))
    safe = (margin > 1e-3).numpy()
    assert np.array_equal(idx.numpy()[safe], ref_idx[safe])
    assert (idx.numpy() == ref_idx).mean() > 0.995
    rs = torch.stack(recon)[:, :, ::4, ::8, ::8]
    np.testing.assert_allclose(rs.numpy(), d["recon_sample"], rtol=2e-3, atol=2e-3)


def test_decode_indices_equals_decode():
    d = load("titok_small.npz")
    sd = seeded_titok_state(0)
    shapes, counts = d["shapes"].tolist(), d["counts"].tolist()
    codes = torch.from_numpy(d["codes"])
    a = O.titok_decode(codes, counts, shapes, sd)
    b = O.titok_decode_indices(torch.from_numpy(d["indices"]), shapes, counts, sd, [7, 5, 5, 5, 5])
    for x, y in zip(a, b):
        assert torch.equal(x, y)


# ---------------------------------------------------------------------------------------------- GAN loss module (section 8f)
def _loss_fixture():
    from titok_video_amd.synthetic import seeded_tower_state
    d = load("loss_kat.npz")
    shapes = [tuple(int(v) for v in s) for s in d["shapes"]]
    target = synthetic_clips(shapes, seed=int(d["clip_seed"]))
    recon = [torch.from_numpy(d[f"recon{i}"]) for i in range(len(shapes))]
    noise = [torch.from_numpy(d[f"noise{i}"]) for i in range(len(shapes))]
    sd = seeded_tower_state("encoder", "tiny", (4, 8, 8), 3, 1, seed=int(d["disc_seed"]))
    return d, target, recon, noise, sd


def test_loss_oracle_matches_reference_loss_module():
    """oracle/loss_oracle.py against the reference's own ReconstructionLoss (fp32, no_grad; make_golden_loss.py)."""
    from oracle import loss_oracle as LO
    d, target, recon, noise, sd = _loss_fixture()
    with torch.no_grad():
        np.testing.assert_allclose(LO.disc_logits(target, sd).numpy(), d["logits_real"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(LO.disc_logits(recon, sd).numpy(), d["logits_fake"], rtol=0, atol=2e-5)
        tot, gd = LO.generator_loss(target, recon, sd, float(d["disc_weight"]))
    assert abs(float(tot) - float(d["gen_total"])) < 1e-5
    for k in ("recon_loss", "g_loss", "total_loss"):
        assert abs(float(gd["gen/" + k]) - float(d["gen_" + k])) < 1e-5, k
    tot, dd = LO.discriminator_loss(target, recon, sd, float(d["gp_weight"]), float(d["gp_noise"]), float(d["centering_weight"]), noise)
    assert abs(float(tot) - float(d["disc_total"])) < 2e-4      # the R1/R2 terms are differences of nearly equal logits times 1/noise^2
    for k in ("d_loss", "logits_relative", "r1_penalty", "r2_penalty", "centering_loss", "total_loss"):
        assert abs(float(dd["disc/" + k]) - float(d["disc_" + k])) < 2e-4, k


def test_loss_oracle_gradients_follow_the_reference_bf16_run():
    """fp32 autograd through the oracle vs the reference's bf16 backward (its real precision): same direction and scale."""
    from oracle import loss_oracle as LO
    d, target, recon, noise, sd = _loss_fixture()
    rec = [r.clone().requires_grad_(True) for r in recon]
    tot, _ = LO.generator_loss(target, rec, sd, float(d["disc_weight"]))
    tot.backward()
    for i, r in enumerate(rec):
        ref = torch.from_numpy(d[f"gen_drecon{i}_bf16"]).double().flatten()
        got = r.grad.double().flatten()
        cos = float(torch.dot(ref, got) / (ref.norm() * got.norm()))
        assert cos > 0.97, (i, cos)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    tot, _ = LO.discriminator_loss(target, recon, sdg, float(d["gp_weight"]), float(d["gp_noise"]), float(d["centering_weight"]), noise)
    tot.backward()
    for n in ("model_layers.attn_layer.0.to_qkv.weight", "model_layers.ffd_layer.3.w3.weight", "proj_out.weight"):
        ref = torch.from_numpy(d["disc_grad_bf16::" + n]).double().flatten()
        got = sdg[n].grad.double().flatten()
        cos = float(torch.dot(ref, got) / (ref.norm() * got.norm()))
        assert cos > 0.9, (n, cos)


@pytest.mark.parametrize("size", ["small", "base", "large"])
def test_oracle_other_sizes_equal_the_reference(size):
    """tests/golden/titok_sizes.npz (reference modules, fp32 and bf16, make_golden_sizes.py): the oracle's fp32 encoder + FSQ gives the
    reference's indices on all 384 tokens and its pre-rounding values to 2e-5 for get_model_dims small / base / large."""
    from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
    g = np.load(os.path.join(G, "titok_sizes.npz"))
    sd = seeded_titok_state(int(g["weight_seed"]), encoder_size=size, decoder_size=size, gain=float(g["weight_gain"]))
    clips = synthetic_clips(g["shapes"].tolist(), seed=int(g["clip_seed"]))
    with torch.no_grad():
        _r, idx, _z, b = O.titok_forward(clips, g["counts"].tolist(), sd, g["levels"].tolist(), size, size)
    assert np.array_equal(idx.numpy(), g[f"{size}_indices"])
    assert float(np.abs(b.numpy() - g[f"{size}_bounded"]).max()) < 2e-5
    # and the reference's own bf16 run is the yardstick the GPU tests use: it differs from its fp32 run
    assert int((g[f"{size}_indices_refbf16"] != g[f"{size}_indices"]).sum()) > 0


def test_oracle_sampling_extremes_equal_the_reference():
    """tests/golden/titok_extremes.npz: the corners of the loader's sampling ranges (largest grid with K = 128 and K = 1) through the
    reference's tiny model - the oracle reproduces its fp32 indices (386 tokens)."""
    from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
    g = np.load(os.path.join(G, "titok_extremes.npz"))
    sd = seeded_titok_state(int(g["weight_seed"]))
    clips = synthetic_clips(g["shapes"].tolist(), seed=int(g["clip_seed"]))
    with torch.no_grad():
        _r, idx, _z, b = O.titok_forward(clips, g["counts"].tolist(), sd, g["levels"].tolist())
    assert np.array_equal(idx.numpy(), g["indices"])
    assert float(np.abs(b.numpy() - g["bounded"]).max()) < 5e-5      # fp32 summation order over 1892-row sequences
