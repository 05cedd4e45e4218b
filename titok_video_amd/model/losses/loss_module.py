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


import os

_F32_HEAD = os.environ.get("TTV_DISC_F32_HEAD", "0") == "1"
_TWO_CALLS = os.environ.get("TTV_DISC_TWO_CALLS", "0") == "1"


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

    # ---- discriminator access -------------------------------------------------------------------------------------------
    def disc_wrapper(self, x: Sequence[torch.Tensor]) -> torch.Tensor:
        """One logit per clip: the mean of the clip's 4 register-token outputs (loss_module.py:96-101)."""
        n = len(x)
        if _F32_HEAD:    # the tower's fp32 token outputs, averaged in fp32: no bf16 rounding of the logits themselves
            per_token = self.disc_model.forward_z(list(x), [self.disc_tokens] * n)  # fp32 [4 n, 1]
        else:            # the reference's dtype flow (encoder output cast to the clips' dtype, blocks.py:103)
            per_token = self.disc_model(list(x), [self.disc_tokens] * n)          # [4 n, 1]
        return per_token.view(n, -1).mean(dim=-1)

    def _set_disc_trainable(self, flag: bool) -> None:
        params = self.__dict__.get("_disc_param_list")          # cached: parameters() walks the module tree, twice per step here
        if params is None:
            params = self.__dict__["_disc_param_list"] = list(self.disc_model.parameters())
        for p in params:
            p.requires_grad = flag

    @staticmethod
    def _report(prefix: str, terms) -> dict:
        """The reference's logging dictionary: every term reduced to a detached scalar under 'gen/..' or 'disc/..'."""
        return {f"{prefix}/{name}": value.clone().mean().detach() for name, value in terms.items()}

    def forward(self, target, recon, disc_forward: bool = False, gp_noise_tensors: Optional[List[torch.Tensor]] = None):
        if disc_forward:
            return self._discriminator_step_loss(target, recon, gp_noise_tensors)
        return self._generator_step_loss(target, recon)

    # ---- generator step (loss_module.py:110-162, perceptual terms off) ----------------------------------------------------
    def _generator_step_loss(self, target, recon):
        real = [t.contiguous() for t in target]
        fake = [r.contiguous() for r in recon]
        # mean over clips of the per-clip L1 means (loss_module.py:118; value and gradient in one HIP launch).  The reference keeps
        # the [B] vector until the final .mean(); mean(a + w b) = mean(a) + w mean(b), so the scalar is carried instead.
        terms = {"recon_loss": l1_reconstruction_loss(fake, real)}
        total = terms["recon_loss"]
        if self.disc_weight > 0.0:
            self._set_disc_trainable(False)                                   # the generator sees a frozen critic (:144-146)
            score_real = self.disc_wrapper([t.detach() for t in real])        # no gradient path: runs the fused inference towers
            score_fake = self.disc_wrapper(fake)                              # tape + inputs-only backward into the reconstruction
            terms["g_loss"] = F.softplus(score_real - score_fake)             # softplus(-(fake - real)), relativistic (:149-151)
            total = total + self.disc_weight * terms["g_loss"].mean()
        terms["total_loss"] = total
        return total, self._report("gen", terms)

    # ---- discriminator step (loss_module.py:165-213) ----------------------------------------------------------------------
    def _discriminator_step_loss(self, target, recon, noise=None):
        # upstream marks both lists requires_grad (:168-169) for an autograd penalty it does not take on this path: the finite-difference
        # R1 / R2 below read logits only, so nothing reads d loss / d clip - without the flag the tower's backward skips its input
        # gradients (patch-embed dX for every packed clip) and autograd keeps no per-clip .grad.  Loss and parameter gradients are the same.
        real = [t.detach().contiguous() for t in target]
        fake = [r.detach().contiguous() for r in recon]
        self._set_disc_trainable(True)
        use_penalty = self.gp_weight > 0.0
        # The reference makes 2 (+2 with the penalty) discriminator calls; clips are independent inside the tower (block-diagonal
        # attention, per-row norms), so they are issued here as ONE packed call and the logits split afterwards: same values,
        # one tape / one backward / one set of weight-gradient launches instead of four.
        packed = real + fake
        if use_penalty:
            if noise is None:
                # one generator call for the whole batch instead of one per clip (a launch each), split into per-clip views
                flat = torch.randn(sum(t.numel() for t in real), dtype=real[0].dtype, device=real[0].device) * self.gp_noise
                noise, off = [], 0
                for t in real:
                    noise.append(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()
            # multi-tensor adds: two launches instead of two per clip
            noisy = list(torch._foreach_add(real, list(noise))) + list(torch._foreach_add(fake, list(noise)))
        if use_penalty and _TWO_CALLS:
            # two packed calls with IDENTICAL plans: clip j and its noisy copy sit at the same packed rows of their call
            scores = torch.cat([self.disc_wrapper(packed), self.disc_wrapper(noisy)]).view(-1, len(real))
        else:
            if use_penalty:
                packed = packed + noisy
            scores = self.disc_wrapper(packed).view(-1, len(real))            # rows: real, fake, (real + noise, fake + noise)
        score_real, score_fake = scores[0], scores[1]
        margin = score_real - score_fake
        terms = {"d_loss": F.softplus(-margin), "logits_relative": margin}
        total = terms["d_loss"]
        if use_penalty:                                                       # finite-difference R1 / R2 (:187-198)
            terms["r1_penalty"] = (score_real - scores[2]).square()
            terms["r2_penalty"] = (score_fake - scores[3]).square()
            total = total + (self.gp_weight / self.gp_noise ** 2) * (terms["r1_penalty"] + terms["r2_penalty"])
        if self.centering_weight > 0.0:                                       # keeps real / fake logits centred on zero (:201-204)
            terms["centering_loss"] = 0.5 * (score_real + score_fake).square()
            total = total + self.centering_weight * terms["centering_loss"]
        total = total.mean()
        terms["total_loss"] = total
        return total, self._report("disc", terms)
