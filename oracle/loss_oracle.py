"""CPU restatement of the reference's GAN loss module for the generator/discriminator steps (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.

Follows model/losses/loss_module.py of the reference with `perceptual_weight = gram_weight = 0` (LPIPS needs weights
fetched from the network):
  * disc_logits            loss_module.py:96-101  (discriminator = TiTokEncoder(out_channels=1), K = 4 register tokens per
                                                   clip, logit = mean over the clip's tokens)
  * generator_loss         loss_module.py:110-162 (per-clip L1 mean + disc_weight * softplus(-(fake - real)), mean over clips)
  * discriminator_loss     loss_module.py:165-213 (softplus(-(real - fake)) + gp_weight/gp_noise^2 * (R1 + R2 finite-difference
                                                   penalties) + centering_weight * (real + fake)^2 / 2, mean over clips)
Pinned by tests/golden/loss_kat.npz (the reference's own ReconstructionLoss run by tests/golden/make_golden_loss.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import titok_oracle as O

DISC_TOKENS = 4   # loss_module.py:42


def disc_logits(clips: Sequence[Tensor], disc_sd: Dict[str, Tensor], model_size: str = "tiny", patch=(4, 8, 8),
                prefix: str = "") -> Tensor:
    """[B] logits: encoder with out_channels = 1 on K = 4 tokens per clip, mean over the tokens (loss_module.py:96-101)."""
    b = len(clips)
    z = O.encoder_forward(list(clips), [DISC_TOKENS] * b, disc_sd, model_size, tuple(patch), prefix=prefix)
    return z.to(clips[0].dtype).view(b, -1).mean(-1)


def generator_loss(target: Sequence[Tensor], recon: Sequence[Tensor], disc_sd: Optional[Dict[str, Tensor]], disc_weight: float,
                   model_size: str = "tiny", patch=(4, 8, 8)) -> Tuple[Tensor, Dict[str, Tensor]]:
    """loss_module.py:110-162 with the perceptual terms off."""
    recon_loss = torch.stack([(x - y).abs().mean() for x, y in zip(target, recon)])   # :118
    out = {"recon_loss": recon_loss}
    g_loss = 0.0
    if disc_weight > 0.0:
        frozen = {k: v.detach() for k, v in disc_sd.items()}                          # :144-146 (requires_grad = False)
        logits_real = disc_logits([t.detach() for t in target], frozen, model_size, patch)
        logits_fake = disc_logits(recon, frozen, model_size, patch)
        g_loss = F.softplus(-(logits_fake - logits_real))                             # :149-151
        out["g_loss"] = g_loss
    total = (recon_loss + disc_weight * g_loss).mean()                                # :155-160
    out["total_loss"] = total
    return total, {"gen/" + k: v.clone().mean().detach() for k, v in out.items()}


def discriminator_loss(target: Sequence[Tensor], recon: Sequence[Tensor], disc_sd: Dict[str, Tensor], gp_weight: float,
                       gp_noise: float, centering_weight: float, noise: Optional[List[Tensor]] = None,
                       model_size: str = "tiny", patch=(4, 8, 8)) -> Tuple[Tensor, Dict[str, Tensor]]:
    """loss_module.py:165-213.  `noise` = the per-clip N(0,1) * gp_noise tensors of :188 (drawn here when None)."""
    target = [t.detach().requires_grad_(True) for t in target]                        # :168-169
    recon = [r.detach().requires_grad_(True) for r in recon]
    logits_real = disc_logits(target, disc_sd, model_size, patch)
    logits_fake = disc_logits(recon, disc_sd, model_size, patch)
    logits_relative = logits_real - logits_fake
    d_loss = F.softplus(-logits_relative)                                             # :178-179
    out = {"d_loss": d_loss, "logits_relative": logits_relative}
    penalty = 0.0
    if gp_weight > 0.0:                                                               # :187-198
        if noise is None:
            noise = [torch.randn_like(x) * gp_noise for x in target]
        real_n = disc_logits([x + n for x, n in zip(target, noise)], disc_sd, model_size, patch)
        fake_n = disc_logits([x + n for x, n in zip(recon, noise)], disc_sd, model_size, patch)
        r1, r2 = (logits_real - real_n) ** 2, (logits_fake - fake_n) ** 2
        out["r1_penalty"], out["r2_penalty"] = r1, r2
        penalty = r1 + r2
    centering = 0.0
    if centering_weight > 0.0:                                                        # :201-204
        centering = ((logits_real + logits_fake) ** 2) / 2
        out["centering_loss"] = centering
    total = (d_loss + (gp_weight / gp_noise ** 2 * penalty) + centering_weight * centering).mean()   # :207-211
    out["total_loss"] = total
    return total, {"disc/" + k: v.clone().mean().detach() for k, v in out.items()}
