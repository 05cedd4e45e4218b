"""HipAdamW (titok_video_amd/optim.py, csrc/ttv_train.hip k_opt_gradsq / k_opt_adamw) against torch.nn.utils.clip_grad_norm_ +
torch.optim.AdamW on the same parameters and gradients - the optimizer step of the reference's training loop (train.py:76-77, :183-190)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
SHAPES = [(256,), (1,), (257,), (64, 129), (8193,), (1408, 256), (3, 5, 7), (16384,), (8192,)]
HYPER = dict(lr=1e-3, betas=(0.5, 0.96), eps=1e-8, weight_decay=1e-2)


def _params(dtype, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(s, generator=g) * 0.5).to(DEV, dtype)) for s in SHAPES]


def _grads(dtype, seed, scale):
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    return [(torch.randn(s, generator=g) * scale).to(DEV, dtype) for s in SHAPES]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("max_norm", [None, 1.0, 1e6])
def test_matches_torch_adamw_over_several_steps(dtype, max_norm):
    from titok_video_amd.optim import HipAdamW
    ours, ref = _params(dtype, 0), _params(dtype, 0)
    opt = HipAdamW(ours, **HYPER)
    opt_ref = torch.optim.AdamW(ref, foreach=False, fused=False, **HYPER)
    for step in range(5):
        gs = _grads(dtype, step, 0.02 * (step + 1))          # total norm ~ 2 .. 12: max_norm = 1 clips, 1e6 does not
        for p, r, g in zip(ours, ref, gs):
            p.grad, r.grad = g.clone(), g.clone()
        if max_norm is None:
            opt.step()
        else:
            norm = opt.clip_and_step(max_norm)
            norm_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm)
            assert abs(float(norm) - float(norm_ref)) <= (1e-5 if dtype == torch.float32 else 1e-2) * float(norm_ref)
            for p, g in zip(ours, gs):
                assert torch.equal(p.grad, g), "HipAdamW must not rewrite p.grad"
        opt_ref.step()
        for i, (p, r) in enumerate(zip(ours, ref)):
            d = (p.detach().float() - r.detach().float()).abs().max().item()
            # fp32: rounding order only; bf16: torch rounds the clipped gradient and the norm to bf16 first, this path keeps fp32 - one bf16 ulp of the
            # parameter (2^-8 relative) at most, on a few elements
            tol = 2e-6 if dtype == torch.float32 else 2.0 ** -7 * max(1e-3, r.detach().float().abs().max().item())
            assert d <= tol, (step, i, tuple(p.shape), d)
    for p, r in zip(ours, ref):
        so, sr = opt.state[p], opt_ref.state[r]
        assert float(so["step"]) == float(sr["step"]) == 5.0
        scale = sr["exp_avg"].float().abs().max().item()
        assert (so["exp_avg"].float() - sr["exp_avg"].float()).abs().max().item() <= (1e-6 if dtype == torch.float32 else 2e-2) * max(scale, 1e-12)


def test_state_dict_round_trip_with_torch_adamw():
    """The state layout is torch's: a HipAdamW state_dict continues in torch.optim.AdamW and the other way round, same trajectory."""
    from titok_video_amd.optim import HipAdamW
    a, b = _params(torch.float32, 3), _params(torch.float32, 3)
    oa, ob = HipAdamW(a, **HYPER), torch.optim.AdamW(b, foreach=False, fused=False, **HYPER)
    for step in range(2):
        for p, r, g in zip(a, b, _grads(torch.float32, step, 0.05)):
            p.grad, r.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    # swap the optimizers' states
    sa, sb = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())
    oa2, ob2 = HipAdamW(a, **HYPER), torch.optim.AdamW(b, foreach=False, fused=False, **HYPER)
    oa2.load_state_dict(sb); ob2.load_state_dict(sa)
    for step in range(2, 4):
        for p, r, g in zip(a, b, _grads(torch.float32, step, 0.05)):
            p.grad, r.grad = g.clone(), g.clone()
        oa2.step(); ob2.step()
    for p, r in zip(a, b):
        assert (p.detach() - r.detach()).abs().max().item() <= 2e-6
        assert float(oa2.state[p]["step"]) == float(ob2.state[r]["step"]) == 4.0


def test_parameters_without_gradient_are_left_alone_and_two_groups_share_one_norm():
    from titok_video_amd.optim import HipAdamW
    ours, ref = _params(torch.float32, 7), _params(torch.float32, 7)
    groups = lambda ps: [dict(params=ps[:4], lr=1e-3), dict(params=ps[4:], lr=3e-3, weight_decay=0.0)]
    opt = HipAdamW(groups(ours), **HYPER)
    opt_ref = torch.optim.AdamW(groups(ref), foreach=False, fused=False, **HYPER)
    gs = _grads(torch.float32, 0, 0.1)
    for i, (p, r, g) in enumerate(zip(ours, ref, gs)):
        if i != 2 and i != 6:
            p.grad, r.grad = g.clone(), g.clone()
    before = [p.detach().clone() for p in ours]
    norm = opt.clip_and_step(0.5)
    norm_ref = torch.nn.utils.clip_grad_norm_([r for r in ref if r.grad is not None], 0.5)
    opt_ref.step()
    assert abs(float(norm) - float(norm_ref)) <= 1e-5 * float(norm_ref)
    for i, (p, r) in enumerate(zip(ours, ref)):
        assert (p.detach() - r.detach()).abs().max().item() <= 2e-6, i
    assert torch.equal(ours[2].detach(), before[2]) and torch.equal(ours[6].detach(), before[6])
    assert not opt.state[ours[2]]


def test_make_optimizer_picks_the_hip_step_and_training_step_uses_it():
    """train.make_optimizer() returns HipAdamW for a model on the GPU and train.training_step() steps through clip_and_step(): the
    parameters after a step are those of clip_grad_norm_(1.0) + torch.optim.AdamW applied to the SAME gradients (p.grad still holds them,
    unclipped; comparing two whole training runs instead would compare the sign of summation noise on the zero-gradient elements)."""
    from types import SimpleNamespace
    from titok_video_amd.model.titok import TiTok
    from titok_video_amd.optim import HipAdamW
    from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
    from titok_video_amd.train import make_optimizer, training_step
    cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny")))
    shapes, counts = [(4, 16, 16), (8, 32, 48)], [2, 5]
    m = TiTok(cfg); m.load_state_dict(seeded_titok_state(0)); m = m.to(DEV, torch.float32).train()
    clips = synthetic_clips(shapes, seed=3, dtype=torch.float32, device=DEV)
    params = [p for p in m.parameters() if p.requires_grad]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt = make_optimizer(m)
    assert isinstance(opt, HipAdamW)
    opt_ref = torch.optim.AdamW(ref, lr=1e-4, betas=(0.5, 0.96), weight_decay=1e-4, foreach=False, fused=False)
    for step in range(3):
        loss, gnorm, _ = training_step(m, clips, counts, opt)
        for p, r in zip(params, ref):
            r.grad = None if p.grad is None else p.grad.detach().clone()
        with_grad = [r for r in ref if r.grad is not None]
        gnorm_ref = torch.nn.utils.clip_grad_norm_(with_grad, 1.0)
        opt_ref.step()
        assert abs(float(gnorm) - float(gnorm_ref)) <= 1e-5 * float(gnorm_ref)
        for p, r in zip(params, ref):
            assert (p.detach() - r.detach()).abs().max().item() <= 2e-6 * max(1.0, r.detach().abs().max().item()), step
            r.data.copy_(p.detach())          # same starting point for the next step
