"""Mirror of the reference's `ReconstructionLoss` (model/losses/loss_module.py) for the L1 + GAN terms.

The discriminator is this package's `TiTokEncoder(out_channels=1)` called with K = 4 register tokens per clip
(loss_module.py:40-48,96-101): 6 of the 8 tower forwards of a reference training step are these calls, and the generator
step differentiates THROUGH the frozen discriminator into the reconstruction (loss_module.py:144-151) — both run on the HIP
path (tape forward + hand-written backward, input-clip gradients included).  Same constructor argument (the config tree),
same `forward(target, recon, disc_forward=False)` signature, same return value `(total_loss, {'gen/..' | 'disc/..': scalar})`
and the same state-dict keys (`disc_model.*`).

Not built: the LPIPS / Gram terms (loss_module.py:28-36,61-94,121-138).  Their VGG weights are fetched from the network by the
reference, so `perceptual_weight` and `gram_weight` must be 0 here; anything else raises.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..base.blocks import TiTokEncoder
from ..base.utils import init_weights
from ...train import l1_reconstruction_loss


class ReconstructionLoss(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        loss_c = config.tokenizer.losses
        loss_d = config.discriminator.losses
        self.perceptual_weight = float(loss_c.perceptual_weight)
        self.gram_weight = float(loss_c.gram_weight)
        if self.perceptual_weight > 0.0 or self.gram_weight > 0.0:
            raise NotImplementedError("the LPIPS / Gram terms need network-fetched VGG weights (reference lpips_gram.py:10-48); "
                                      "set tokenizer.losses.perceptual_weight = gram_weight = 0 on this path")
        model_d = config.discriminator.model
        self.disc_weight = float(loss_c.disc_weight)
        if self.disc_weight > 0.0:
            self.disc_tokens = 4   # extra as register tokens (loss_module.py:42)
            self.disc_model = TiTokEncoder(model_size=model_d.model_size, patch_size=tuple(model_d.patch_size), in_channels=3,
                                           out_channels=1).apply(init_weights)
        self.gp_weight = float(loss_d.gp_weight)
        self.gp_noise = float(loss_d.gp_noise)
        self.centering_weight = float(loss_d.centering_weight)
        self.total_steps = config.training.main.max_steps

    def disc_wrapper(self, x: Sequence[torch.Tensor]) -> torch.Tensor:
        """[B] logits = mean over each clip's 4 register tokens (loss_module.py:96-101)."""
        b = len(x)
        logits = self.disc_model(list(x), [self.disc_tokens] * b).view(b, -1).mean(-1)
        return logits

    def forward(self, target, recon, disc_forward: bool = False, gp_noise_tensors: Optional[List[torch.Tensor]] = None):
        if disc_forward:
            return self._forward_discriminator(target, recon, gp_noise_tensors)
        return self._forward_generator(target, recon)

    def _forward_generator(self, target, recon):
        loss_dict = {}
        target = [i.contiguous() for i in target]
        recon = [i.contiguous() for i in recon]
        # mean over clips of the per-clip L1 means (loss_module.py:118; value and gradient in one HIP launch).  The reference keeps
        # the [B] vector until the final .mean(); mean(a + w b) = mean(a) + w mean(b), so the scalar is carried instead.
        recon_loss = l1_reconstruction_loss(recon, target)
        loss_dict["recon_loss"] = recon_loss
        g_loss = 0.0
        if self.disc_weight > 0.0:
            target = [i.detach().contiguous() for i in target]
            for param in self.disc_model.parameters():                 # loss_module.py:144-146
                param.requires_grad = False
            logits_real = self.disc_wrapper(target)
            logits_fake = self.disc_wrapper(recon)
            logits_relative = logits_fake - logits_real
            g_loss = F.softplus(-logits_relative)
            loss_dict["g_loss"] = g_loss
        total_loss = recon_loss + (self.disc_weight * g_loss.mean() if self.disc_weight > 0.0 else 0.0)
        loss_dict["total_loss"] = total_loss
        return total_loss, {"gen/" + k: v.clone().mean().detach() for k, v in loss_dict.items()}

    def _forward_discriminator(self, target, recon, noise=None):
        loss_dict = {}
        target = [i.detach().requires_grad_(True).contiguous() for i in target]
        recon = [i.detach().requires_grad_(True).contiguous() for i in recon]
        for param in self.disc_model.parameters():                     # loss_module.py:172-174
            param.requires_grad = True
        # The reference makes 2 (+2 with the penalty) discriminator calls; clips are independent inside the tower (block-diagonal
        # attention, per-row norms), so they are issued here as ONE packed call and the logits split afterwards: same values,
        # one tape / one backward / one set of weight-gradient launches instead of four.
        b = len(target)
        batch = list(target) + list(recon)
        if self.gp_weight > 0.0:                                       # finite-difference R1 / R2 (loss_module.py:187-198)
            if noise is None:
                noise = [torch.randn_like(x) * self.gp_noise for x in target]
            batch += [x + y for x, y in zip(target, noise)] + [x + y for x, y in zip(recon, noise)]
        logits = self.disc_wrapper(batch)
        logits_real, logits_fake = logits[:b], logits[b:2 * b]
        logits_relative = logits_real - logits_fake
        d_loss = F.softplus(-logits_relative)
        loss_dict["d_loss"] = d_loss
        loss_dict["logits_relative"] = logits_relative
        gradient_penalty = 0.0
        if self.gp_weight > 0.0:
            logits_real_noised, logits_fake_noised = logits[2 * b:3 * b], logits[3 * b:]
            r1_penalty = (logits_real - logits_real_noised) ** 2
            r2_penalty = (logits_fake - logits_fake_noised) ** 2
            loss_dict["r1_penalty"] = r1_penalty
            loss_dict["r2_penalty"] = r2_penalty
            gradient_penalty = r1_penalty + r2_penalty
        centering_loss = 0.0
        if self.centering_weight > 0.0:
            centering_loss = ((logits_real + logits_fake) ** 2) / 2
            loss_dict["centering_loss"] = centering_loss
        total_loss = (d_loss + (self.gp_weight / self.gp_noise ** 2 * gradient_penalty)
                      + (self.centering_weight * centering_loss)).mean()
        loss_dict["total_loss"] = total_loss
        return total_loss, {"disc/" + k: v.clone().mean().detach() for k, v in loss_dict.items()}
