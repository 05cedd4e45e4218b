"""Golden fixture for the GAN loss module: runs the REFERENCE's own ReconstructionLoss (model/losses/loss_module.py).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_loss.py

The module imports torchvision and model.metrics.lpips_gram (LPIPS, network-fetched VGG weights) at the top; both are absent /
unusable offline and are replaced before import by inert stand-ins (never called: perceptual_weight = gram_weight = 0).
flash_attn / xformers are replaced as in make_golden.py.  The discriminator weights come from the seed recipe
(titok_video_amd.synthetic.seeded_tower_state), so the fixture holds inputs / outputs only:
  * fp32, no_grad (the reference cannot run fp32 with grad): generator and discriminator loss dictionaries,
    logits, with the R1/R2 noise captured
  * bf16 with grad: d(generator loss)/d(recon) and the discriminator step's parameter-gradient norms (yardstick for the bf16 path)
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets sys.path for the reference and this repo)


def install_loss_standins():
    import torch.nn as nn
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    v2 = types.ModuleType("torchvision.transforms.v2")
    v2.functional = types.SimpleNamespace(resize=None)
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvf.InterpolationMode = types.SimpleNamespace(BICUBIC="bicubic")
    tvt.v2 = v2
    tvt.functional = tvf
    tv.transforms = tvt
    lp = types.ModuleType("model.metrics.lpips_gram")
    lp.LPIPS = type("LPIPS", (nn.Module,), {})
    for name, mod in [("torchvision", tv), ("torchvision.transforms", tvt), ("torchvision.transforms.v2", v2),
                      ("torchvision.transforms.functional", tvf), ("model.metrics.lpips_gram", lp)]:
        sys.modules[name] = mod


def loss_config(disc_weight=0.4, gp_weight=0.1, gp_noise=0.1, centering_weight=0.01):
    return SimpleNamespace(
        tokenizer=SimpleNamespace(losses=SimpleNamespace(disc_weight=disc_weight, perceptual_weight=0.0, gram_weight=0.0,
                                                         perceptual_samples_per_step=24, perceptual_sampling_size=128)),
        discriminator=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], model_size="tiny"),
                                      losses=SimpleNamespace(gp_weight=gp_weight, gp_noise=gp_noise, centering_weight=centering_weight)),
        training=SimpleNamespace(main=SimpleNamespace(torch_compile=False, max_steps=1000)))


SHAPES = [(4, 16, 16), (8, 32, 48), (4, 8, 24)]
DISC_SEED = 77


def main():
    torch.set_num_threads(8)
    MG.install_standins()
    install_loss_standins()
    from titok_video_amd.synthetic import seeded_tower_state, synthetic_clips
    from model.losses.loss_module import ReconstructionLoss

    cfg = loss_config()
    mod = ReconstructionLoss(cfg)
    sd = seeded_tower_state("encoder", "tiny", (4, 8, 8), 3, 1, seed=DISC_SEED)
    mod.disc_model.load_state_dict(sd, strict=True)
    target = synthetic_clips(SHAPES, seed=21)
    g = torch.Generator().manual_seed(5)
    recon = [t * 0.8 + 0.1 * torch.randn(t.shape, generator=g) for t in target]
    noise = [torch.randn(t.shape, generator=g) * cfg.discriminator.losses.gp_noise for t in target]

    out = {"disc_seed": np.int64(DISC_SEED), "clip_seed": np.int64(21), "shapes": np.array(SHAPES, dtype=np.int32),
           "disc_weight": np.float64(0.4), "gp_weight": np.float64(0.1), "gp_noise": np.float64(0.1), "centering_weight": np.float64(0.01)}
    for i, (r, n) in enumerate(zip(recon, noise)):
        out[f"recon{i}"] = MG.np32(r)
        out[f"noise{i}"] = MG.np32(n)

    # ---- fp32 values (no_grad) ----
    with torch.no_grad():
        out["logits_real"] = MG.np32(mod.disc_wrapper(target))
        out["logits_fake"] = MG.np32(mod.disc_wrapper(recon))
        tot, d = mod(target, recon)
        out["gen_total"] = MG.np32(tot)
        for k, v in d.items():
            out["gen_" + k.split("/")[1]] = MG.np32(v)
    # the discriminator branch calls requires_grad_ on its inputs and draws the R1/R2 noise with randn_like: run it with grad
    # disabled for the arithmetic, feeding the captured noise through a patched randn_like
    it = iter(noise)
    real_randn_like = torch.randn_like
    torch.randn_like = lambda x, **kw: next(it) / cfg.discriminator.losses.gp_noise
    try:
        with torch.no_grad():
            tot, d = mod(target, recon, disc_forward=True)
    finally:
        torch.randn_like = real_randn_like
    out["disc_total"] = MG.np32(tot)
    for k, v in d.items():
        out["disc_" + k.split("/")[1]] = MG.np32(v)

    # ---- bf16 gradients (the reference's real precision; grad works in bf16) ----
    modb = ReconstructionLoss(cfg)
    modb.disc_model.load_state_dict(sd, strict=True)
    modb = modb.to(torch.bfloat16)
    tb = [t.to(torch.bfloat16) for t in target]
    rb = [r.to(torch.bfloat16).requires_grad_(True) for r in recon]
    tot, _ = modb(tb, rb)
    tot.backward()
    out["gen_total_bf16"] = MG.np32(tot)
    for i, r in enumerate(rb):
        out[f"gen_drecon{i}_bf16"] = MG.np32(r.grad)
    it = iter(noise)
    torch.randn_like = lambda x, **kw: (next(it) / cfg.discriminator.losses.gp_noise).to(x.dtype)
    try:
        modb.zero_grad(set_to_none=True)
        tot, _ = modb(tb, [r.detach() for r in rb], disc_forward=True)
        tot.backward()
    finally:
        torch.randn_like = real_randn_like
    out["disc_total_bf16"] = MG.np32(tot)
    names, norms = [], []
    for n, p in modb.disc_model.named_parameters():
        names.append(n)
        norms.append(float(p.grad.float().norm()))
    out["disc_grad_names"] = np.array(names)
    out["disc_grad_norms_bf16"] = np.array(norms, dtype=np.float64)
    big = dict(modb.disc_model.named_parameters())
    for n in ("model_layers.attn_layer.0.to_qkv.weight", "model_layers.ffd_layer.3.w3.weight", "proj_out.weight"):
        out["disc_grad_bf16::" + n] = MG.np32(big[n].grad)
    MG.save("loss_kat.npz", **out)


if __name__ == "__main__":
    main()
