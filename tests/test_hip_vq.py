"""Nearest-codebook-entry (L2) quantiser (csrc/ttv_vq.hip) - BASELINE.json north_star / configs #4, #5.  NOT a reference component
(the reference quantises with FSQ only), so it is pinned through FSQ: on FSQ's own lattice it must return FSQ's indices - the
reference-generated fsq_kat.npz fixture - and on synthetic codebooks it must equal the float64 cdist + argmin oracle
(oracle/vq_oracle.py; lowest index on ties) wherever the two best distances are further apart than fp32 rounding."""
import os

import numpy as np
import pytest
import torch

from oracle import titok_oracle as O
from oracle import vq_oracle as V

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_lattice_argmin_is_fsq(tag):
    """CPU: nearest lattice entry of the bounded vector == the reference's FSQ index (fixture from the unmodified fsq.py), except
    where a bounded value sits at a rounding boundary."""
    d = np.load(os.path.join(G, "fsq_kat.npz"))
    levels = d[f"levels_{tag}"].tolist()
    b = torch.from_numpy(d[f"bounded_{tag}"])
    lat = V.fsq_lattice(levels)
    assert lat.shape == (int(np.prod(levels)), len(levels))
    idx, best, gap = V.l2_argmin(b, lat)
    safe = O.fsq_margin(b) > 1e-4
    assert int(safe.sum()) > 0.9 * len(safe)
    assert torch.equal(idx[safe], torch.from_numpy(d[f"indices_{tag}"])[safe])
    # lattice entries are the reference's implicit codebook times levels // 2
    hw = torch.tensor(levels) // 2
    assert torch.equal(lat, torch.from_numpy(d[f"codebook_{tag}"]) * hw)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_lattice_argmin_is_fsq(tag):
    from titok_video_amd.model.quantizer.fsq import FSQ
    from titok_video_amd.model.quantizer.vq_l2 import L2Quantizer
    d = np.load(os.path.join(G, "fsq_kat.npz"))
    levels = d[f"levels_{tag}"].tolist()
    f = FSQ(levels)
    z = torch.from_numpy(d[f"z_{tag}"]).to(DEV)
    vq = L2Quantizer(f.lattice_codebook()).to(DEV)
    assert torch.equal(vq.codebook.detach().cpu(), V.fsq_lattice(levels))
    bounded = f.bounded(z)
    idx = vq.indices(bounded).cpu()
    ref_b = torch.from_numpy(d[f"bounded_{tag}"])
    safe = O.fsq_margin(ref_b) > 1e-4
    assert torch.equal(idx[safe], torch.from_numpy(d[f"indices_{tag}"])[safe])
    # and on the model's own FSQ kernel, token for token
    _codes, dd = f(z)
    assert torch.equal(idx[safe], dd["indices"].cpu()[safe])
    # straight-through lookup returns lattice entries; scaled back they are FSQ's codes
    q = vq.lookup(idx.to(DEV))
    hw = (torch.tensor(levels) // 2).float().to(DEV)
    assert torch.equal((q / hw).cpu()[safe], torch.from_numpy(d[f"codes_{tag}"])[safe])


@pytest.mark.gpu
@pytest.mark.parametrize("N,C", [(8192, 32), (16384, 64), (1000, 7), (100, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_l2_argmin_matches_cdist_oracle(N, C, dtype):
    """Synthetic codebooks of configs #4 / #5 (8192 x 32, 16384 x 64) plus ragged sizes (N not a multiple of the 128-entry chunk,
    odd C).  Inputs are rounded to `dtype` first, so the oracle sees exactly what the kernel sees; indices must agree wherever the
    runner-up is further away than the accumulated fp32 rounding (gap > 1e-4 x the best distance's scale), and on EVERY row the
    kernel's pick must be within that rounding of the true minimum."""
    from titok_video_amd.model.quantizer.vq_l2 import L2Quantizer
    g = torch.Generator().manual_seed(N + C)
    cb = torch.randn(N, C, generator=g).to(dtype)
    rows = 3000
    z = (torch.randn(rows, C, generator=g) * 1.2).to(dtype)
    z[:64] = cb[torch.randint(0, N, (64,), generator=g)]            # exact hits: distance 0
    if N >= 1000:
        cb[777] = cb[12]                                            # an exact duplicate: the lower index must win
        z[64:72] = cb[12]
    vq = L2Quantizer(cb.float()).to(DEV)
    idx, dist = vq.indices(z.to(DEV), want_distance=True)
    idx, dist = idx.cpu(), dist.cpu().double()
    ref_idx, ref_best, gap = V.l2_argmin(z.float(), cb.float())
    scale = (z.double().pow(2).sum(1) + ref_best + 1.0)
    tol = 2e-5 * scale if dtype == torch.float32 else 2e-5 * scale
    safe = gap > tol
    assert float(safe.float().mean()) > (0.97 if C > 1 else 0.85)     # one dimension, 100 entries: many near-ties
    assert torch.equal(idx[safe], ref_idx[safe]), int((idx[safe] != ref_idx[safe]).sum())
    picked = (z.double() - cb.double()[idx.long()]).pow(2).sum(1)
    assert bool((picked <= ref_best + tol).all())
    assert float((dist - picked).abs().max()) < 1e-3 * float(scale.max())
    # exact hits: the entry itself, unless a neighbour is closer than fp32 can resolve (dense one-dimensional codebooks)
    assert bool((idx[:64] == ref_idx[:64])[safe[:64]].all()) and bool((picked[:64] <= tol[:64]).all())
    if N >= 1000:
        assert bool((idx[64:72] == 12).all())


@pytest.mark.gpu
def test_l2_quantizer_module_straight_through():
    from titok_video_amd.model.quantizer.vq_l2 import L2Quantizer
    g = torch.Generator().manual_seed(3)
    vq = L2Quantizer(torch.randn(512, 32, generator=g)).to(DEV)
    z = torch.randn(200, 32, generator=g).to(DEV).requires_grad_(True)
    codes, info = vq(z)
    assert info["indices"].dtype == torch.int32 and codes.shape == z.shape
    # z + (q - z) in fp32: equal to the entry up to one rounding
    assert torch.allclose(codes.detach(), vq.codebook.detach()[info["indices"].long()], rtol=0, atol=1e-6)
    w = torch.randn_like(codes)
    (codes * w).sum().backward()
    assert torch.equal(z.grad, w)                      # d codes / d z = identity (straight-through)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(8192, 32), (1024, 64)])
def test_titok_with_l2_quantizer_wired_in(dtype, shape):
    """`quantizer: l2` (build-defined, NOT reference-pinned: the reference ships FSQ only, titok.py:37): encode -> nearest codebook
    entry -> straight-through lookup -> decode through `TiTok.forward`, token widths 32 / 64 (the widths of BASELINE configs #4 / #5),
    against the oracle towers + the cdist oracle.  float32 towers: every index whose runner-up is further than the fp32 rounding of the
    towers is the oracle's, pixels follow; bf16: the decoder is judged on the indices the HIP encoder produced, and the share of
    indices equal to the fp32 oracle's is reported against what a bf16 execution of the oracle towers gets."""
    from types import SimpleNamespace
    from oracle import titok_oracle as O
    from titok_video_amd.model.titok import TiTok
    from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
    n_entries, width = shape
    cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(
        patch_size=[4, 8, 8], fsq_levels=None, quantizer="l2", codebook_size=n_entries, token_size=width, encoder_size="tiny", decoder_size="tiny")))
    sd = seeded_titok_state(3, token_size=width)
    model = TiTok(cfg)
    cb = torch.randn(n_entries, width, generator=torch.Generator().manual_seed(9)) * 1.5
    model.load_state_dict({**sd, "quantize.codebook": cb}, strict=True)
    shapes, counts = [(8, 32, 48), (4, 16, 16), (16, 64, 64)], [17, 1, 128]
    cpu_clips = synthetic_clips(shapes, seed=5)
    with torch.no_grad():
        z_ref = O.encoder_forward(cpu_clips, counts, sd, "tiny", prefix="encoder.")
    idx_ref, _, gap = V.l2_argmin(z_ref, cb)
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    model = model.to(DEV, tdt).eval()
    clips = [c.to(DEV, tdt) for c in cpu_clips]
    with torch.no_grad():
        recon, info = model(clips, counts)
        codes, info2 = model.encode(clips, counts, split_indices=True)
    idx = info["indices"].cpu()
    assert torch.equal(torch.cat(list(info2["indices"])).cpu(), idx) and codes.shape == (sum(counts), width)
    if dtype == "f32":
        safe = gap > 1e-3 * (1.0 + z_ref.double().pow(2).sum(1).sqrt())      # runner-up beyond the towers' fp32 rounding
        assert int(safe.sum()) > 0.9 * idx.numel()
        assert torch.equal(idx[safe], idx_ref[safe])
    else:
        with torch.no_grad():    # the bf16 yardstick: the oracle towers run in bf16 on the same inputs
            z_y = O.encoder_forward([c.to(torch.bfloat16) for c in cpu_clips], counts, sd, "tiny", prefix="encoder.")
        idx_y, _, _ = V.l2_argmin(z_y.float(), cb)
        agree, agree_y = float((idx == idx_ref).float().mean()), float((idx_y == idx_ref).float().mean())
        print(f"l2 {shape}: bf16 HIP indices equal to the fp32 oracle's {agree:.3f}, bf16 oracle towers {agree_y:.3f}")
        assert agree >= agree_y - 0.08
    # decoder on the indices the HIP encoder produced (lookup of the same codebook rows)
    with torch.no_grad():
        dec_ref = O.decoder_forward(cb[idx.long()], counts, shapes, sd, "tiny", prefix="decoder.")
        again = model.decode_indices(info["indices"], shapes, counts)
    tol = 5e-3
    if dtype == "bf16":      # yardstick: the oracle decoder run in bf16 on the same codes (fixed factor 1.5, floor 0.25)
        with torch.no_grad():
            dec_y = O.decoder_forward(cb[idx.long()].to(torch.bfloat16), counts, shapes, sd, "tiny", prefix="decoder.")
        tol = max(0.25, 1.5 * max(float((y.float() - ref).abs().max()) for y, ref in zip(dec_y, dec_ref)))
    for r, a, ref in zip(recon, again, dec_ref):
        assert float((r.float().cpu() - ref).abs().max()) < tol
        assert torch.equal(r, a)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1024, 64), (512, 32)])
def test_training_through_the_wired_l2_quantizer_fp32(shape):
    """VERDICT round 3, missing #1: the reference wires exactly one quantiser and trains through it (titok.py:37,47-52, train.py:65-83).
    `quantizer: l2` under autograd: tape-recording towers with token widths 32 / 64 (the d <-> token projections' backward in column
    chunks), straight-through towards the encoder, scatter-add of the decoder's token gradient into the selected codebook rows.
    fp32 gradients of every tower parameter, the codebook and the input clips against torch autograd over the oracle towers with the
    same estimator; tolerance 2e-3 relative (Frobenius) as in tests/test_hip_backward.py."""
    from types import SimpleNamespace
    from oracle import titok_oracle as O
    from titok_video_amd.model.titok import TiTok
    from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
    n_entries, width = shape
    cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(
        patch_size=[4, 8, 8], fsq_levels=None, quantizer="l2", codebook_size=n_entries, token_size=width, encoder_size="tiny", decoder_size="tiny")))
    sd0 = seeded_titok_state(3, token_size=width)
    cb0 = torch.randn(n_entries, width, generator=torch.Generator().manual_seed(9)) * 1.5
    shapes, counts = [(4, 16, 16), (8, 32, 48), (4, 8, 24)], [2, 9, 3]
    # ---- oracle: autograd through the CPU towers, lookup with the same straight-through estimator
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    cb = cb0.clone().requires_grad_(True)
    ref_clips = [c.requires_grad_(True) for c in synthetic_clips(shapes, seed=21)]
    target = [c.detach() * 0.5 for c in ref_clips]
    z = O.encoder_forward(ref_clips, counts, sd, "tiny", prefix="encoder.")
    idx_ref, _, gap = V.l2_argmin(z.detach(), cb0)
    codes = cb[idx_ref.long()] + (z - z.detach())
    recon = O.decoder_forward(codes, counts, shapes, sd, "tiny", prefix="decoder.")
    loss_ref = sum((r - t).abs().mean() for r, t in zip(recon, target)) + 0.1 * z.pow(2).mean()
    loss_ref.backward()
    # ---- HIP
    model = TiTok(cfg)
    model.load_state_dict({**sd0, "quantize.codebook": cb0}, strict=True)
    model = model.to(DEV, torch.float32).train()
    clips = [c.requires_grad_(True) for c in synthetic_clips(shapes, seed=21, dtype=torch.float32, device=DEV)]
    codes_h, info = model.encode(clips, counts, [tuple(c.shape[1:]) for c in clips])
    recon_h = model.decode(codes_h, counts, [tuple(c.shape[1:]) for c in clips])
    z_h = model.encoder.forward_z(clips, counts)          # the regulariser's z (a second tape; its gradient adds to the first)
    loss = sum((r.float() - t.to(DEV)).abs().mean() for r, t in zip(recon_h, target)) + 0.1 * z_h.pow(2).mean()
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(info["indices"].cpu(), idx_ref), "token indices differ (a runner-up within fp32 rounding would be a test-data issue)"
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * abs(float(loss_ref))

    def rel(a, b):
        a, b = a.double().cpu(), b.double().cpu()
        return float((a - b).norm() / (b.norm() + 1e-30))
    worst = 0.0
    for name, p in model.named_parameters():
        ref = cb.grad if name == "quantize.codebook" else sd[name].grad
        assert p.grad is not None, name
        e = rel(p.grad, ref)
        worst = max(worst, e)
        assert e < 2e-3, (name, e)
    assert int((cb.grad.abs().sum(1) > 0).sum()) == int(idx_ref.unique().numel())     # exactly the selected rows receive gradient
    for c, rc in zip(clips, ref_clips):
        assert rel(c.grad, rc.grad) < 2e-3
    print(f"l2 {shape} fp32 training step: worst relative gradient error {worst:.2e}")
