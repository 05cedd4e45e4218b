"""Host-side tables and init for the towers.

Mirrors the interface of the reference's model/base/utils.py (get_model_dims 8-23, init_weights
54-66).  The reference's einops patch/unpatch helpers (26-51) have no host counterpart here: the
rearrangement runs inside the proj_in / proj_out GEMM kernels (csrc/ttv_gemm.hip: k_gemm_bf16<STORE, GATHER>, k_gemm_k256<STORE_PATCH>)
or, for other patch shapes / dtypes, in k_patch_copy (csrc/ttv_elem.hip).
"""
from __future__ import annotations

import torch
import torch.nn as nn

_LAYERS = {"tiny": 4, "small": 8, "base": 12, "large": 24}
_HEADS = {"tiny": [4, 2], "small": [8, 2], "base": [12, 4], "large": [16, 4]}


def get_model_dims(model_size: str = "tiny", head_dim: int = 64, mlp_ratio: float = 4.0):
    """(width, layers, [q_heads, kv_heads], mlp_ratio) - same table as reference utils.py:8-23."""
    if model_size not in _LAYERS:
        raise KeyError(model_size)
    heads = list(_HEADS[model_size])
    return int(head_dim * heads[0]), _LAYERS[model_size], heads, mlp_ratio


def geglu_inner_dim(dim: int, mult: float = 4.0, mult_of: int = 32) -> int:
    """GEGLU hidden width (reference transformer.py:39-40): int(mult*2/3*dim) rounded up to 32."""
    inner = int(mult * (2 / 3) * dim)
    return mult_of * ((inner + mult_of - 1) // mult_of)


class RMSNormWeight(nn.Module):
    """Parameter holder with the state-dict shape of the reference's RMSNorm (`weight [dim]`, no bias).

    The normalisation itself is never run by this module: the HIP kernels read `weight`.
    """

    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = None


class LinearWeight(nn.Module):
    """Parameter holder with the state-dict shape of `nn.Linear` (`weight [out,in]`, optional bias)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        nn.init.trunc_normal_(self.weight, mean=0.0, std=0.02)


def init_weights(module: nn.Module) -> None:
    """Reference init (utils.py:54-66): linear weights trunc_normal(0, 0.02), biases 0, norm gains 1."""
    if isinstance(module, (nn.Linear, LinearWeight)):
        nn.init.trunc_normal_(module.weight.data, mean=0.0, std=0.02)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)
    elif isinstance(module, (nn.LayerNorm, RMSNormWeight)):
        if getattr(module, "bias", None) is not None:
            nn.init.constant_(module.bias, 0)
        if module.weight is not None:
            nn.init.constant_(module.weight, 1.0)
    # `.data` writes above do not bump the parameters' version counters: a tower drops its packed weight copies explicitly
    # (model.apply(init_weights) visits the towers too)
    if hasattr(module, "invalidate_packs"):
        module.invalidate_packs()
