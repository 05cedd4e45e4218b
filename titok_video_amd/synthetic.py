"""Seeded synthetic inputs for measurement and parity runs.

There is no network for checkpoints or datasets, so benchmarks and parity tests use
* clips  : `[3,T,H,W]`, values U(-1,1) (the loader's range, reference dataset/video_dataset.py:118-119)
* weights: a deterministic recipe keyed by the reference's state-dict names.  The reference's own
  init (trunc_normal std 0.02, model/base/utils.py:54-58) maps every latent token to a single FSQ
  index, so the recipe scales the linear weights up and perturbs the norm gains to
  get non-degenerate token indices.

The recipe only depends on torch's CPU generator, so it reproduces bit-identically on any box with
this image; golden fixtures therefore store inputs/outputs only, never weights.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import torch

from .model.base.utils import geglu_inner_dim, get_model_dims


def tower_param_shapes(kind: str, model_size: str, patch_size: Sequence[int], in_channels: int,
                       out_channels: int) -> "OrderedDict[str, Tuple[int, ...]]":
    """State-dict keys and shapes of one tower (reference model/base/blocks.py:31-69,108-146)."""
    width, layers, heads, _ = get_model_dims(model_size)
    hq, hkv = heads
    g = (width // hq) * hkv
    inner = geglu_inner_dim(width)
    pd = math.prod(patch_size)
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    if kind == "encoder":
        s["proj_in.weight"] = (width, in_channels * pd)
        s["proj_in.bias"] = (width,)
    else:
        s["proj_in.weight"] = (width, in_channels)
        s["proj_in.bias"] = (width,)
    s["mask_token"] = (1, 1)
    s["ln_pre_t.weight"] = (width,)
    s["ln_pre_p.weight"] = (width,)
    for i in range(layers):
        s[f"model_layers.attn_layer.{i}.pre_ln.weight"] = (width,)
        s[f"model_layers.attn_layer.{i}.to_qkv.weight"] = (2 * width + 2 * g, width)
        s[f"model_layers.attn_layer.{i}.out_proj.weight"] = (width, width)
    for i in range(layers):
        s[f"model_layers.ffd_layer.{i}.norm.weight"] = (width,)
        s[f"model_layers.ffd_layer.{i}.w12.weight"] = (2 * inner, width)
        s[f"model_layers.ffd_layer.{i}.w3.weight"] = (width, inner)
    for i in range(layers - 1):
        s[f"model_layers.attn_post_ln.{i}.weight"] = (width,)
    for i in range(layers - 1):
        s[f"model_layers.ffd_post_ln.{i}.weight"] = (width,)
    s["ln_post.weight"] = (width,)
    if kind == "encoder":
        s["proj_out.weight"] = (out_channels, width)
        s["proj_out.bias"] = (out_channels,)
    else:
        s["proj_out.weight"] = (out_channels * pd, width)
        s["proj_out.bias"] = (out_channels * pd,)
    return s


def seeded_tower_state(kind: str, model_size: str, patch_size: Sequence[int], in_channels: int,
                       out_channels: int, seed: int, gain: float = 6.0) -> Dict[str, torch.Tensor]:
    """fp32 CPU state dict for one tower from `seed` (keys in `tower_param_shapes` order)."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(seed))
    out: Dict[str, torch.Tensor] = OrderedDict()
    for key, shape in tower_param_shapes(kind, model_size, patch_size, in_channels, out_channels).items():
        r = torch.randn(shape, generator=gen, dtype=torch.float32)
        if key == "mask_token":
            t = 0.5 * r
        elif key.endswith("bias"):
            t = 0.02 * r
        elif len(shape) == 1:          # RMSNorm gains
            t = 1.0 + 0.1 * r
        else:                          # Linear weights
            t = (0.02 * gain) * r.clamp(-2.0, 2.0)
        out[key] = t.contiguous()
    return out


def seeded_titok_state(seed: int = 0, encoder_size: str = "tiny", decoder_size: str = "tiny",
                       patch_size: Sequence[int] = (4, 8, 8), token_size: int = 5,
                       gain: float = 6.0) -> Dict[str, torch.Tensor]:
    """State dict of a whole `TiTok` (`encoder.*`, `decoder.*`; FSQ has no persistent keys)."""
    sd: Dict[str, torch.Tensor] = OrderedDict()
    enc = seeded_tower_state("encoder", encoder_size, patch_size, 3, token_size, 1000 + seed, gain)
    dec = seeded_tower_state("decoder", decoder_size, patch_size, token_size, 3, 2000 + seed, gain)
    for k, v in enc.items():
        sd["encoder." + k] = v
    for k, v in dec.items():
        sd["decoder." + k] = v
    return sd


def synthetic_clips(shapes: Sequence[Sequence[int]], seed: int = 1234,
                    dtype: torch.dtype = torch.float32, device="cpu") -> List[torch.Tensor]:
    """List of `[3,T,H,W]` clips, U(-1,1), generated on CPU from `seed` then moved/cast."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(seed))
    clips = []
    for (t, h, w) in shapes:
        c = torch.rand((3, int(t), int(h), int(w)), generator=gen, dtype=torch.float32) * 2.0 - 1.0
        clips.append(c.to(device=device, dtype=dtype))
    return clips


def synthetic_token_counts(n: int, lo: int, hi: int, seed: int = 1234) -> List[int]:
    """K ~ U[lo, hi] per clip (reference configs/tiny.yaml:57 token_range)."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(seed) + 7)
    return torch.randint(lo, hi + 1, (n,), generator=gen).tolist()
