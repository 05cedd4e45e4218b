"""Checkpoint compatibility with the reference trainer.

The reference saves Lightning checkpoints of `TitokTrainer` (train.py:28-37): a dict whose 'state_dict' holds the tokenizer under
`model.` and the discriminator under `loss_module.disc_model.`, with metric / LPIPS entries filtered out (train.py:218-220);
`init_from_checkpoint` loads it with strict=False (train.py:265-267).  The mirrors keep the reference's parameter names, so these
helpers only add / strip the trainer prefixes.  Pure host code (no GPU needed).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

MODEL_PREFIX = "model."
LOSS_PREFIX = "loss_module."
_SKIP = ("eval_metrics", "perceptual_model")      # never saved by the reference (train.py:218-220)


def trainer_state_dict(model: torch.nn.Module, loss_module: Optional[torch.nn.Module] = None) -> "OrderedDict[str, torch.Tensor]":
    """What `TitokTrainer.state_dict()` returns for these two modules."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in model.state_dict().items():
        out[MODEL_PREFIX + k] = v
    if loss_module is not None:
        for k, v in loss_module.state_dict().items():
            if not any(s in k for s in _SKIP):
                out[LOSS_PREFIX + k] = v
    return out


def split_trainer_state_dict(state_dict: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """(tokenizer entries, loss-module entries) with the trainer prefixes removed; metric / LPIPS entries dropped."""
    model_sd, loss_sd = OrderedDict(), OrderedDict()
    for k, v in state_dict.items():
        if any(s in k for s in _SKIP):
            continue
        if k.startswith(MODEL_PREFIX):
            model_sd[k[len(MODEL_PREFIX):]] = v
        elif k.startswith(LOSS_PREFIX):
            loss_sd[k[len(LOSS_PREFIX):]] = v
    return model_sd, loss_sd


def load_trainer_state_dict(state_dict: Dict[str, torch.Tensor], model: torch.nn.Module, loss_module: Optional[torch.nn.Module] = None,
                            strict: bool = True):
    """Load a reference trainer state dict into the mirrors.  A bare tokenizer state dict (no prefixes) is accepted too."""
    model_sd, loss_sd = split_trainer_state_dict(state_dict)
    if not model_sd and not loss_sd:
        model_sd = state_dict
    res = [model.load_state_dict(model_sd, strict=strict)]
    if loss_module is not None and (loss_sd or strict):
        res.append(loss_module.load_state_dict(loss_sd, strict=strict))
    return res


def save_checkpoint(path: str, model: torch.nn.Module, loss_module: Optional[torch.nn.Module] = None, global_step: int = 0) -> None:
    """Write the subset of a Lightning checkpoint the reference reads back (train.py:265-267: ['state_dict'], 'global_step')."""
    sd = OrderedDict((k, v.detach().cpu()) for k, v in trainer_state_dict(model, loss_module).items())
    torch.save({"state_dict": sd, "global_step": int(global_step)}, path)


def load_checkpoint(path: str, model: torch.nn.Module, loss_module: Optional[torch.nn.Module] = None, strict: bool = False) -> int:
    """`init_from_checkpoint` (train.py:265-267; the reference loads with strict=False).  Returns the stored global step."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    load_trainer_state_dict(ck["state_dict"], model, loss_module, strict=strict)
    return int(ck.get("global_step", 0))
