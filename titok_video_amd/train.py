"""Generator training step on the HIP path (mirrors reference train.py:65-83 with the reconstruction term of
model/losses/loss_module.py:118: L1 between target and reconstruction, mean over clips of per-clip means).

    forward (tape) -> L1 loss -> backward (HIP kernels) -> [DP: count-weighted gradient all-reduce over RCCL]
    -> clip_grad_norm(max_grad_norm) -> optimizer.step()

LPIPS and the GAN discriminator are outside this path's scope (they need network-fetched weights);
the discriminator's tower itself is `TiTokEncoder(out_channels=1)` and differentiates through its inputs on this path.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import dp


def l1_reconstruction_loss(recon: Sequence[torch.Tensor], target: Sequence[torch.Tensor]) -> torch.Tensor:
    """Mean over clips of mean |target - recon| (loss_module.py:118 with per-clip tensors of different shapes)."""
    terms = [(r.float() - t.float()).abs().mean() for r, t in zip(recon, target)]
    return torch.stack(terms).mean()


def make_optimizer(model: torch.nn.Module, lr: float = 1e-4, beta1: float = 0.5, beta2: float = 0.96, weight_decay: float = 1e-4):
    """AdamW with the reference's hyper-parameters (configs/tiny.yaml:39-46, train.py:170-190)."""
    return torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=lr, betas=(beta1, beta2), weight_decay=weight_decay)


def training_step(model, clips: List[torch.Tensor], token_counts, optimizer, max_grad_norm: float = 1.0,
                  target: Optional[List[torch.Tensor]] = None, group=None):
    """One generator step on this rank's clips.  Returns (loss, grad_norm, indices)."""
    optimizer.zero_grad(set_to_none=True)
    recon, out = model(clips, token_counts)
    loss = l1_reconstruction_loss(recon, target if target is not None else clips)
    loss.backward()
    params = [p for p in model.parameters() if p.grad is not None]
    dp.allreduce_mean_by_count([p.grad for p in params], len(clips), group=group)
    gnorm = torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
    optimizer.step()
    return loss.detach(), gnorm, out["indices"]
