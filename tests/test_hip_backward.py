"""Training-step parity: gradients of the HIP towers (tape forward + hand-written backward
kernels) against torch autograd through the CPU oracle on the same seeded inputs.  `-m gpu`.

  * fp32 compute: every parameter gradient and the input-clip gradients within 2e-3 relative (Frobenius) of the oracle's.
  * bf16 compute: per-parameter cosine similarity >= 0.97 and norm ratio within 12 % of the fp32 oracle gradients
    (bf16 activations/gradients, fp32 accumulation; the straight-through FSQ makes tokens near a rounding boundary
    legitimately differ, so the fixture uses the decoder-side loss on the reference codes for the tight part).
  * single kernels (weight-gradient GEMM, RMSNorm backward, attention backward) against torch autograd directly.
"""
import ctypes as C
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
LEVELS = [7, 5, 5, 5, 5]
DT = {"bf16": torch.bfloat16, "f32": torch.float32}


def L():
    return _lib.lib()


def S():
    return _lib.stream_ptr(torch.device(DEV))


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def config():
    return SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(
        patch_size=[4, 8, 8], fsq_levels=LEVELS, encoder_size="tiny", decoder_size="tiny")))


# ---------------------------------------------------------------------------------------------- single kernels
@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("shape", [(1000, 256, 768), (333, 768, 256), (2500, 1408, 256), (130, 256, 704), (36864, 768, 256), (64, 256, 256),
                                   (70, 8, 264), (4097, 136, 120), (1500, 256, 256), (2000, 264, 136), (1090, 128, 128)])   # 3 / 4 / 3 (last one ragged) stages per block: the ring's edges
@pytest.mark.parametrize("workspace", [False, True])
def test_wgrad(dt, shape, workspace):
    Lr, N, K = shape
    g = torch.Generator().manual_seed(Lr)
    dy = torch.randn(Lr, N, generator=g).to(DT[dt])
    x = torch.randn(Lr, K, generator=g).to(DT[dt])
    dyd, xd = dy.to(DEV), x.to(DEV)
    nbytes = int(L().ttv_linear_wgrad_workspace_bytes(Lr, N, K)) if workspace else 0
    ws = torch.full((max(nbytes, 4) // 4,), float("nan"), device=DEV)   # stale partials must never leak into dW
    outs = []
    for _ in range(2):     # two independent runs: with a workspace the split sum has a fixed order
        dw = torch.zeros(N, K, device=DEV)
        for _ in range(2):     # accumulates
            _lib.check(L().ttv_linear_wgrad(dyd.data_ptr(), N, xd.data_ptr(), K, dw.data_ptr(), K, Lr, N, K, _lib.dtype_code(DT[dt]),
                                            ws.data_ptr() if workspace else None, nbytes, S()), "wgrad")
        outs.append(dw)
    ref = 2 * (dy.double().T @ x.double())
    assert rel(outs[0], ref) < (2e-3 if dt == "bf16" else 2e-5)
    if workspace and dt == "bf16":     # the fp32 checking kernel accumulates with atomics
        assert torch.equal(outs[0], outs[1])


def _rms_bwd_ref(x, gain, dy):
    xr = x.double().requires_grad_(True)
    gr = gain.double().requires_grad_(True)
    yy = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * gr
    yy.backward(dy.double())
    return xr.grad, gr.grad


@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("second", [False, True])
@pytest.mark.parametrize("rows,d", [(77, 256), (2051, 256), (33, 512), (9, 1024), (130, 320)])
def test_rmsnorm_backward_chain(dt, second, rows, d):
    g = torch.Generator().manual_seed(rows + d)
    x = (torch.randn(rows, d, generator=g) * 2).to(DT[dt])
    dy = torch.randn(rows, d, generator=g).to(DT[dt])
    acc = torch.randn(rows, d, generator=g)
    y = torch.randn(rows, d, generator=g) * 3
    g1 = 1 + 0.1 * torch.randn(d, generator=g)
    g2 = 1 + 0.1 * torch.randn(d, generator=g)
    alpha = 16.0
    da, dg1_ref = _rms_bwd_ref(x.float(), g1, dy.float())
    h = acc.double() + da
    if second:
        out, dg2_ref = _rms_bwd_ref(y, g2, h)
    else:
        out, dg2_ref = h, None
    dx = acc.to(DEV).clone()
    cast = torch.empty(rows, d, dtype=DT[dt], device=DEV)
    dg1 = torch.zeros(d, device=DEV)
    dg2 = torch.zeros(d, device=DEV)
    xd, dyd, yd, g1d, g2d = x.to(DEV), dy.to(DEV), y.to(DEV), g1.to(DEV), g2.to(DEV)
    _lib.check(L().ttv_rmsnorm_backward_chain(xd.data_ptr(), d, dyd.data_ptr(), d, g1d.data_ptr(), dg1.data_ptr(), dx.data_ptr(), d,
                                              yd.data_ptr() if second else None, d, g2d.data_ptr() if second else None,
                                              dg2.data_ptr() if second else None, alpha if second else 1.0, cast.data_ptr(), d, rows, d, 1e-5,
                                              _lib.dtype_code(DT[dt]), S()), "chain")
    scale = alpha if second else 1.0
    assert rel(dx, scale * out) < 1e-5
    assert rel(cast, out) < (4e-3 if dt == "bf16" else 1e-5)
    assert rel(dg1, dg1_ref) < 1e-5
    if second:
        assert rel(dg2, dg2_ref) < 1e-5


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_rmsnorm_backward(dt):
    rows, d = 77, 256
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(rows, d, generator=g) * 2).to(DT[dt])
    dy = torch.randn(rows, d, generator=g).to(DT[dt])
    gain = 1 + 0.1 * torch.randn(d, generator=g)
    xr = x.double().requires_grad_(True)
    gr = gain.double().requires_grad_(True)
    y = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * gr
    y.backward(dy.double())
    dx = torch.empty(rows, d, dtype=DT[dt], device=DEV)
    dg = torch.zeros(d, device=DEV)
    xd, dyd, gd = x.to(DEV), dy.to(DEV), gain.to(DEV)
    _lib.check(L().ttv_rmsnorm_backward(xd.data_ptr(), d, dyd.data_ptr(), d, gd.data_ptr(), dx.data_ptr(), d, dg.data_ptr(), rows, d, 1e-5,
                                        _lib.dtype_code(DT[dt]), S()), "rmsnorm_bwd")
    assert rel(dx.float(), xr.grad) < (6e-3 if dt == "bf16" else 1e-5)
    assert rel(dg, gr.grad) < (1e-3 if dt == "bf16" else 1e-5)


@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("case", [([(4, 16, 16)], [3]), ([(8, 32, 48), (4, 8, 24), (8, 32, 32)], [5, 3, 70])])
@pytest.mark.parametrize("rope", [False, True])
def test_attention_backward(dt, case, rope):   # the LSE forward below runs with half items too (plan's default for small batches)
    shapes, counts = case
    plan = BatchPlan(shapes, counts, (4, 8, 8), DEV)
    hq, hkv, d, gq = 4, 2, 256, 128
    ld = 2 * d + 2 * gq
    Lr = plan.total_rows
    g = torch.Generator().manual_seed(7)
    qkvg = torch.randn(Lr, ld, generator=g).to(DT[dt])
    dout = torch.randn(Lr, d, generator=g).to(DT[dt])
    # reference: autograd through the oracle's attention
    f = qkvg.double().requires_grad_(True)
    q, gt, k, v = f.split([d, d, gq, gq], dim=-1)
    ref_o = O.attention_varlen(q.unflatten(-1, (hq, 64)), k.unflatten(-1, (hkv, 64)), v.unflatten(-1, (hkv, 64)), plan.cu_seqlens).flatten(-2)
    ref_o.backward(dout.double())
    code = _lib.dtype_code(DT[dt])
    xd, dod = qkvg.to(DEV), dout.to(DEV)
    o = torch.empty(Lr, d, dtype=DT[dt], device=DEV)
    lse = torch.empty(Lr, hq, device=DEV)
    tab = plan.attention_table(hq, hkv)
    _lib.check(L().ttv_attention_lse(xd.data_ptr(), ld, o.data_ptr(), d, plan.cu_dev.data_ptr(), tab.data_ptr(), tab.shape[0], hq, hkv, 64, 0,
                                     code, lse.data_ptr(), S()), "attention_lse")
    # LSE check
    with torch.no_grad():
        qq = qkvg.double()[:, :d].unflatten(-1, (hq, 64)); kk = qkvg.double()[:, 2 * d:2 * d + gq].unflatten(-1, (hkv, 64))
        cu = plan.cu_seqlens
        ref_lse = torch.empty(Lr, hq, dtype=torch.float64)
        for b in range(len(cu) - 1):
            sc = torch.einsum("qhd,khd->hqk", qq[cu[b]:cu[b + 1]], kk[cu[b]:cu[b + 1]].repeat_interleave(hq // hkv, 1)) * 0.125
            ref_lse[cu[b]:cu[b + 1]] = torch.logsumexp(sc, -1).T
    assert float((lse.cpu().double() - ref_lse).abs().max()) < (2e-2 if dt == "bf16" else 1e-4)
    dq = torch.zeros(Lr, ld, dtype=DT[dt], device=DEV)
    delta = torch.empty(Lr, hq, device=DEV)
    scratch = torch.empty(Lr, 2 * gq, device=DEV)
    bt = plan.table(4, 2 * plan.n_blocks64)
    rs = plan.table(5, Lr)
    _lib.check(L().ttv_attention_backward(xd.data_ptr(), ld, o.data_ptr(), d, dod.data_ptr(), d, lse.data_ptr(), delta.data_ptr(),
                                          plan.cu_dev.data_ptr(), bt.data_ptr(), plan.n_blocks64, rs.data_ptr(), dq.data_ptr(), ld,
                                          scratch.data_ptr(), Lr, hq, hkv, code, plan.rope_cs.data_ptr() if rope else None, S()),
               "attention_backward")
    gref = f.grad.clone()
    if rope:   # dq, dk come back as gradients w.r.t. the UN-rotated q, k: the transposed rotation of the plain ones
        cs = plan.rope_cs.cpu().double()
        cos, sin = cs[:, :32].unsqueeze(1), cs[:, 32:].unsqueeze(1)

        def unrotate(gpart):
            gh = gpart.unflatten(-1, (-1, 32, 2))
            return torch.stack((gh[..., 0] * cos + gh[..., 1] * sin, gh[..., 1] * cos - gh[..., 0] * sin), -1).flatten(-3)
        gref[:, :d] = unrotate(gref[:, :d])
        gref[:, 2 * d:2 * d + gq] = unrotate(gref[:, 2 * d:2 * d + gq])
    tol = 2.5e-2 if dt == "bf16" else 2e-4
    assert rel(dq[:, :d].float(), gref[:, :d]) < tol                     # dq
    assert rel(dq[:, 2 * d:2 * d + gq].float(), gref[:, 2 * d:2 * d + gq]) < tol   # dk
    assert rel(dq[:, 2 * d + gq:].float(), gref[:, 2 * d + gq:]) < tol             # dv


# ---------------------------------------------------------------------------------------------- whole training step
def _reference_grads(shapes, counts, seed, use_clip_grad=True):
    sd = {k: v.clone().requires_grad_(True) for k, v in seeded_titok_state(0).items()}
    clips = [c.requires_grad_(use_clip_grad) for c in synthetic_clips(shapes, seed=seed)]
    target = [c.detach() * 0.5 for c in clips]
    recon, idx, z, bounded = O.titok_forward(clips, counts, sd, LEVELS)
    loss = sum((r - t).abs().mean() for r, t in zip(recon, target)) + 0.1 * z.pow(2).mean()
    loss.backward()
    return loss.detach(), idx, sd, clips


def _hip_grads(dtype, shapes, counts, seed, use_clip_grad=True):
    model = TiTok(config())
    model.load_state_dict(seeded_titok_state(0), strict=True)
    model = model.to(DEV, dtype).train()
    clips = [c.requires_grad_(use_clip_grad) for c in synthetic_clips(shapes, seed=seed, dtype=dtype, device=DEV)]
    target = [c.detach() * 0.5 for c in clips]
    z = model.encoder.forward_z(clips, counts)
    codes, dd = model.quantize(z)
    recon = model.decode(codes.to(dtype), counts, [tuple(c.shape[1:]) for c in clips])
    loss = sum((r.float() - t.float()).abs().mean() for r, t in zip(recon, target)) + 0.1 * z.pow(2).mean()
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach(), dd["indices"], model, clips


def test_training_step_gradients_fp32():
    shapes, counts = [(4, 16, 16), (8, 32, 48), (4, 8, 24)], [2, 5, 3]
    ref_loss, ref_idx, sd, ref_clips = _reference_grads(shapes, counts, 31)
    loss, idx, model, clips = _hip_grads(torch.float32, shapes, counts, 31)
    assert torch.equal(idx.cpu(), ref_idx)
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    worst = 0.0
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        e = rel(p.grad, sd[name].grad)
        worst = max(worst, e)
        assert e < 2e-3, (name, e)
    for c, rc in zip(clips, ref_clips):
        assert rel(c.grad, rc.grad) < 2e-3
    print(f"fp32 training step: worst relative gradient error over {len(sd)} parameters: {worst:.2e}")


def _tower_grad_report(model_params, sd, prefix):
    rows, tot_b, tot_d = [], 0.0, 0.0
    for name, p in model_params:
        g, r = p.grad.double().cpu().flatten(), sd[prefix + name].grad.double().flatten()
        cos = float((g @ r) / (g.norm() * r.norm() + 1e-30))
        rows.append((cos, float(g.norm() / (r.norm() + 1e-30)), name, r.numel()))
        tot_d += float((g - r).pow(2).sum()); tot_b += float(r.pow(2).sum())
    rows.sort()
    return rows, float(np.sqrt(tot_d / tot_b))


def test_decoder_gradients_bf16():
    """bf16 tape + backward of the decoder alone (no discrete step in between): codes fixed, L1 loss on the pixels."""
    shapes, counts = [(4, 16, 16), (8, 32, 48), (4, 8, 24)], [2, 5, 3]
    g = torch.Generator().manual_seed(3)
    idx = torch.randint(0, 4375, (sum(counts),), generator=g, dtype=torch.int32)
    codes = O.fsq_indices_to_codes(idx, LEVELS)                                   # exactly representable in bf16
    target = [c * 0.5 for c in synthetic_clips(shapes, seed=9)]
    sd = {k: v.clone().requires_grad_(True) for k, v in seeded_titok_state(0).items()}
    cr = codes.clone().requires_grad_(True)
    recon = O.titok_decode(cr, counts, shapes, sd)
    sum((r - t).abs().mean() for r, t in zip(recon, target)).backward()
    model = TiTok(config()); model.load_state_dict(seeded_titok_state(0)); model = model.to(DEV, torch.bfloat16).train()
    cd = codes.to(DEV, torch.bfloat16).requires_grad_(True)
    rec = model.decode(cd, counts, shapes)
    sum((r.float() - t.to(DEV)).abs().mean() for r, t in zip(rec, target)).backward()
    rows, glob = _tower_grad_report(list(model.decoder.named_parameters()), sd, "decoder.")
    print(f"bf16 decoder backward: global rel. gradient error {glob:.4f}; lowest cosines " + ", ".join(f"{n}={c:.3f}" for c, q, n, _ in rows[:4]))
    assert glob < 0.12      # L1 loss: sign(recon - target) flips for pixels within bf16 noise of the target
    for cos, ratio, name, numel in rows:
        assert cos > (0.98 if numel >= 4096 else 0.93), (name, cos, ratio)
    assert rel(cd.grad.float(), cr.grad) < 0.2          # 50 numbers, each the d -> 5 contraction of a noisy bf16 row gradient


def test_encoder_gradients_bf16():
    """bf16 tape + backward of the encoder alone: smooth loss on the pre-quantisation tokens z; also input-clip gradients
    (the discriminator path differentiates through the encoder's inputs, loss_module.py:149-152)."""
    shapes, counts = [(4, 16, 16), (8, 32, 48), (4, 8, 24)], [2, 5, 3]
    g = torch.Generator().manual_seed(4)
    wz = torch.randn(sum(counts), 5, generator=g)
    sd = {k: v.clone().requires_grad_(True) for k, v in seeded_titok_state(0).items()}
    clips_ref = [c.requires_grad_(True) for c in synthetic_clips(shapes, seed=8)]
    z = O.encoder_forward(clips_ref, counts, sd, "tiny", (4, 8, 8), prefix="encoder.")
    ((z * wz).sum() + 0.1 * z.pow(2).sum()).backward()
    model = TiTok(config()); model.load_state_dict(seeded_titok_state(0)); model = model.to(DEV, torch.bfloat16).train()
    clips = [c.requires_grad_(True) for c in synthetic_clips(shapes, seed=8, dtype=torch.bfloat16, device=DEV)]
    zz = model.encoder.forward_z(clips, counts)
    ((zz * wz.to(DEV)).sum() + 0.1 * zz.pow(2).sum()).backward()
    rows, glob = _tower_grad_report(list(model.encoder.named_parameters()), sd, "encoder.")
    print(f"bf16 encoder backward: global rel. gradient error {glob:.4f}; lowest cosines " + ", ".join(f"{n}={c:.3f}" for c, q, n, _ in rows[:4]))
    assert glob < 0.12
    for cos, ratio, name, numel in rows:
        assert cos > (0.97 if numel >= 4096 else 0.90), (name, cos, ratio)
    for c, rc in zip(clips, clips_ref):
        assert rel(c.grad.float(), rc.grad) < 0.10


def test_training_step_bf16_runs_end_to_end():
    """Whole bf16 training step (encode -> straight-through FSQ -> decode -> L1): loss close to the fp32 oracle and every
    parameter receives a finite gradient.  (Per-parameter agreement is checked per tower above: end to end, bf16 and fp32
    disagree on ~15 % of the discrete token indices, which legitimately changes the decoder's inputs.)"""
    shapes, counts = [(4, 16, 16), (8, 32, 48), (4, 8, 24)], [2, 5, 3]
    ref_loss, ref_idx, sd, ref_clips = _reference_grads(shapes, counts, 31)
    loss, idx, model, clips = _hip_grads(torch.bfloat16, shapes, counts, 31)
    assert abs(float(loss) - float(ref_loss)) < 0.05 * abs(float(ref_loss))
    for name, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        assert p.grad.dtype == p.dtype and p.grad.shape == p.shape
    assert all(c.grad is not None and torch.isfinite(c.grad).all() for c in clips)


def test_training_step_updates_weights_like_reference():
    """One AdamW step (train.py:170-215 hyper-parameters) on the HIP path == the same step on the oracle (fp32)."""
    shapes, counts = [(4, 16, 16), (4, 8, 24)], [2, 3]
    ref_loss, _, sd, _ = _reference_grads(shapes, counts, 5, use_clip_grad=False)
    loss, _, model, _ = _hip_grads(torch.float32, shapes, counts, 5, use_clip_grad=False)
    params_ref = [sd[n] for n, _ in model.named_parameters()]
    opt_ref = torch.optim.AdamW(params_ref, lr=1e-4, betas=(0.5, 0.96), weight_decay=1e-4)
    opt = torch.optim.AdamW(list(model.parameters()), lr=1e-4, betas=(0.5, 0.96), weight_decay=1e-4)
    torch.nn.utils.clip_grad_norm_(params_ref, 1.0)
    torch.nn.utils.clip_grad_norm_(list(model.parameters()), 1.0)
    opt_ref.step(); opt.step()
    for (n, p), r in zip(model.named_parameters(), params_ref):
        assert rel(p.detach(), r.detach()) < 1e-5, n


def test_graphed_training_step_replays_the_eager_step():
    """GraphedTrainingStep: the HIP-graph replay of a training step leaves the parameters where the eager step leaves them (three
    steps each, fresh optimizers, same clips; bf16: the norm-gain gradients accumulate with fp32 atomics, hence a tolerance)."""
    from titok_video_amd.train import GraphedTrainingStep, make_optimizer, training_step
    shapes, counts = [(4, 16, 16), (8, 32, 48), (4, 8, 24)], [2, 5, 3]
    clips = synthetic_clips(shapes, seed=31, dtype=torch.bfloat16, device=DEV)

    def fresh():
        m = TiTok(config())
        m.load_state_dict(seeded_titok_state(0), strict=True)
        return m.to(DEV).train()                                      # fp32 master weights, bf16 compute
    eager, graphed = fresh(), fresh()
    opt_e = make_optimizer(eager, lr=1e-3, capturable=True)
    opt_g = make_optimizer(graphed, lr=1e-3, capturable=True)
    step = GraphedTrainingStep(graphed, opt_g, clips, counts)
    for (n, p), (_, q) in zip(eager.named_parameters(), graphed.named_parameters()):
        assert torch.equal(p, q), n                                   # the capture's warm-up steps were undone
    losses = []
    for _ in range(3):
        le, _, ie = training_step(eager, clips, counts, opt_e)
        lg, _, ig = step(clips)
        losses.append((float(le), float(lg)))
        assert torch.equal(ie, ig)
    for le, lg in losses:
        assert abs(le - lg) < 2e-3 * abs(le), losses
    assert losses[2][0] < losses[0][0]                                # it trains
    for (n, p), (_, q) in zip(eager.named_parameters(), graphed.named_parameters()):
        assert float((p - q).detach().abs().max()) < 2e-3 * max(1.0, float(p.detach().abs().max())), n



_GRAD_DIGEST = r"""
import hashlib, sys, torch
sys.path.insert(0, sys.argv[1])
from tests.test_hip_backward import _hip_grads
shapes, counts = [(4, 16, 16), (8, 32, 48), (4, 8, 24), (16, 64, 64)], [2, 5, 3, 40]
loss, idx, model, clips = _hip_grads(torch.bfloat16, shapes, counts, 31)
h = hashlib.sha256()
for name, p in sorted(model.named_parameters()):
    if any(k in name for k in ("to_qkv", "out_proj", "w12", "w3")) and name.endswith("weight"):
        h.update(name.encode()); h.update(p.grad.float().cpu().numpy().tobytes())
print("DIGEST", h.hexdigest(), float(loss))
"""


def test_weight_gradient_stream_changes_no_bit():
    """The layers' weight-gradient GEMMs run on a second stream beside the dX chain (ttv_train.hip, wgrad_side); the switch
    TTV_WGRAD_SIDE=0 keeps them on the caller's stream.  Same kernels, same split plan, same summing order: the gradients of every
    layer linear must be bit-identical (a missing dependency between the streams would show as a difference).  The switch is read once
    per process and the backward runs on autograd's own thread, so the two settings run in child processes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for side in ("1", "0"):
        env = dict(os.environ, TTV_WGRAD_SIDE=side)
        out = subprocess.run([sys.executable, "-c", _GRAD_DIGEST, root], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([ln for ln in out.stdout.splitlines() if ln.startswith("DIGEST")][0])
    assert digests[0] == digests[1], digests


def test_mixed_precision_training_step_fp32_master_weights_bf16_compute():
    """The reference trains bf16-mixed (fp32 parameters, bf16 autocast).  Here the compute dtype of a tower is the dtype of the
    clips it is given and the weight pack holds compute-dtype copies of the parameters, so fp32 parameters + bf16 clips IS that mode:
    the forward / backward run on the bf16 kernels, the gradients arrive in fp32 on the fp32 parameters, and an update far below
    bf16 resolution is kept by the master weights."""
    from titok_video_amd.train import make_optimizer, training_step
    shapes, counts = [(4, 16, 16), (8, 32, 48), (4, 8, 24)], [2, 5, 3]
    model = TiTok(config())
    model.load_state_dict(seeded_titok_state(0), strict=True)
    model = model.to(DEV).train()                                     # fp32 master weights
    assert all(p.dtype == torch.float32 for p in model.parameters())
    clips = synthetic_clips(shapes, seed=31, dtype=torch.bfloat16, device=DEV)
    # same step with pure-bf16 parameters: the kernels see the same bf16 weight copies, so the gradients must agree closely
    ref = TiTok(config())
    ref.load_state_dict(seeded_titok_state(0), strict=True)
    ref = ref.to(DEV, torch.bfloat16).train()
    idx = []
    for m in (model, ref):
        recon, out = m(clips, counts)
        idx.append(out["indices"].clone())
        assert recon[0].dtype == torch.bfloat16
        from titok_video_amd.train import l1_reconstruction_loss
        l1_reconstruction_loss(recon, [c * 0.5 for c in clips]).backward()
    same_indices = torch.equal(idx[0], idx[1])
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), n
        g, h = p.grad.double().flatten(), q.grad.double().flatten()
        if float(h.norm()) > 0 and p.numel() >= 4096:
            # not bit-equal by construction: the bf16 model rounds norm gains, biases and the gradients themselves to bf16.  A token
            # whose index flips between the two runs (one of ten here; seen when the attention kernels' row maximum was fixed in round 5:
            # 3570 vs 3605) legitimately changes the decoder's input and with it a tenth of the encoder's gradient: then the
            # directions only have to stay related (cf. the per-tower tests above, which hold the indices fixed)
            assert float((g @ h) / (g.norm() * h.norm() + 1e-30)) > (0.9 if same_indices else 0.6), (n, same_indices)
    # one optimizer step with a learning rate whose update (~1e-7 relative) a bf16 parameter could not represent
    for m in (model, ref):
        m.zero_grad(set_to_none=True)
    opt = make_optimizer(model, lr=1e-7, weight_decay=0.0)
    before = [p.detach().clone() for p in model.parameters()]
    loss, gnorm, idx = training_step(model, clips, counts, opt, target=[c * 0.5 for c in clips])
    assert torch.isfinite(loss) and torch.isfinite(gnorm) and idx.numel() == sum(counts)
    moved = sum(int((a != p.detach()).sum()) for a, p in zip(before, model.parameters()))
    total = sum(p.numel() for p in model.parameters())
    assert moved > 0.5 * total, (moved, total)                          # fp32 masters keep the 1e-7 update
    worst = max(float((a - p.detach()).abs().max()) for a, p in zip(before, model.parameters()))
    assert worst < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_l1_loss_matches_autograd(dt):
    """ttv_l1_loss (value + gradient, one launch) vs torch autograd over the per-clip definition (loss_module.py:118)."""
    from titok_video_amd.train import l1_reconstruction_loss
    g = torch.Generator().manual_seed(3)
    shapes = [(3, 4, 16, 16), (3, 8, 32, 48), (3, 4, 8, 24), (3, 5, 7, 3)]    # the last one: 315 elements, vector body + scalar tail
    recon = [torch.randn(sh, generator=g).to(dt) for sh in shapes]
    target = [torch.randn(sh, generator=g).to(dt) for sh in shapes]
    target[0][0, 0, 0, :4] = recon[0][0, 0, 0, :4]                            # exact ties -> zero gradient
    ref_in = [r.clone().float().requires_grad_(True) for r in recon]
    ref = torch.stack([(r - t.float()).abs().mean() for r, t in zip(ref_in, target)]).mean()
    ref.backward()
    dev_in = [r.to("cuda:0").requires_grad_(True) for r in recon]
    loss = l1_reconstruction_loss(dev_in, [t.to("cuda:0") for t in target])
    (loss * 2.0).backward()
    assert abs(float(loss) - float(ref)) < (1e-6 if dt == torch.float32 else 2e-3)
    for a, b in zip(dev_in, ref_in):
        torch.testing.assert_close(a.grad.float().cpu(), 2.0 * b.grad, rtol=1e-2 if dt == torch.bfloat16 else 1e-6, atol=1e-9)
