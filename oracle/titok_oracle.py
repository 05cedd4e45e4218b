"""CPU oracle for the TiTok-Video tokenizer hot path (encode -> FSQ -> decode).

TEST INFRASTRUCTURE ONLY.  This is a plain-PyTorch, functional, CPU restatement of the
reference algorithm.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it; the product path (`titok_video_amd`) never does and fails loudly when
its HIP library is missing.

Every function cites the reference lines (relative to /root/reference) it restates.

Pinning (see tests/golden/make_golden.py and tests/test_oracle_golden.py):
  * FSQ, RoPE table, rotary apply, CodebookLogger: pinned against the reference's own modules,
    imported unmodified (model/quantizer/fsq.py, model/base/rope.py, train_utils/codebook_logging.py).
  * Towers / TiTok forward: pinned against the reference's own `TiTok` imported with local
    stand-ins for the two absent third-party packages (`flash_attn`, `xformers`; versions are
    not pinned anywhere in the reference tree).  At that third-party boundary (varlen attention,
    RMSNorm) parity is anchored on the published definitions restated below - "parity unpinned"
    for flash-attn's own arithmetic, pinned for everything the reference itself implements.

State dicts use the reference's key names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

RMS_EPS = 1e-5  # flash_attn.ops.triton.layer_norm.RMSNorm default eps (blocks.py:27,51-52)
HEAD_DIM = 64   # model/base/utils.py:8


# ----------------------------------------------------------------------------------------------
# model/base/utils.py
# ----------------------------------------------------------------------------------------------
def model_dims(model_size: str) -> Tuple[int, int, Tuple[int, int]]:
    """(width, layers, (q_heads, kv_heads)) - model/base/utils.py:8-23."""
    layers = {"tiny": 4, "small": 8, "base": 12, "large": 24}[model_size]
    heads = {"tiny": (4, 2), "small": (8, 2), "base": (12, 4), "large": (16, 4)}[model_size]
    return HEAD_DIM * heads[0], layers, heads


def geglu_inner_dim(dim: int, mult: float = 4.0, mult_of: int = 32) -> int:
    """model/base/transformer.py:39-40."""
    inner = int(mult * (2 / 3) * dim)
    return mult_of * ((inner + mult_of - 1) // mult_of)


def patchify(clip: Tensor, patch: Sequence[int]) -> Tensor:
    """`c (d0 p0)(d1 p1)(d2 p2) -> (d0 d1 d2)(p0 p1 p2 c)` - model/base/utils.py:26-34."""
    C, T, H, W = clip.shape
    pt, ph, pw = patch
    x = clip.reshape(C, T // pt, pt, H // ph, ph, W // pw, pw)
    x = x.permute(1, 3, 5, 2, 4, 6, 0)  # d0 d1 d2 p0 p1 p2 c
    return x.reshape((T // pt) * (H // ph) * (W // pw), pt * ph * pw * C)


def unpatchify(patches: Tensor, grid: Sequence[int], patch: Sequence[int], channels: int) -> Tensor:
    """`(d0 d1 d2)(p0 p1 p2 c) -> c (d0 p0)(d1 p1)(d2 p2)` - model/base/utils.py:37-51."""
    gt, gh, gw = [int(g) for g in grid]
    pt, ph, pw = patch
    x = patches.reshape(gt, gh, gw, pt, ph, pw, channels)
    x = x.permute(6, 0, 3, 1, 4, 2, 5)  # c d0 p0 d1 p1 d2 p2
    return x.reshape(channels, gt * pt, gh * ph, gw * pw)


# ----------------------------------------------------------------------------------------------
# model/base/rope.py
# ----------------------------------------------------------------------------------------------
def rope_inv_freqs(head_dim: int = HEAD_DIM, grid_dims: int = 3, theta: float = 10000.0) -> Tensor:
    """fp64 `theta ** linspace(0,1,F) * pi/2`, F = head_dim // (2*grid_dims) - rope.py:40-45."""
    n = head_dim // (grid_dims * 2)
    return torch.pow(theta, torch.linspace(0.0, 1.0, n, dtype=torch.float64)) * torch.pi / 2.0


def rope_ids(grid: Sequence[int], token_count: int) -> Tensor:
    """Per-row (t,h,w) position ids for one clip, latent rows first - rope.py:59-67.

    latent i -> (i,i,i); patch (t,h,w) -> (t,h,w) + token_count.  fp32 like the reference.
    """
    k = int(token_count)
    tok = torch.arange(k, dtype=torch.float32).unsqueeze(-1).expand(-1, len(grid))
    coords = [torch.arange(int(g), dtype=torch.float32) for g in grid]
    grid_ids = torch.cartesian_prod(*coords) + k
    if grid_ids.dim() == 1:
        grid_ids = grid_ids.unsqueeze(-1)
    return torch.cat([tok, grid_ids], dim=0)


def rope_table(grids: Sequence[Sequence[int]], token_counts: Sequence[int],
               head_dim: int = HEAD_DIM) -> Tuple[Tensor, Tensor]:
    """(cos, sin) fp64 tables [L, 3F] with column f*3+axis (interleaved) - rope.py:48-54,57-71."""
    ids = torch.cat([rope_ids(g, k) for g, k in zip(grids, token_counts)], dim=0)
    inv = rope_inv_freqs(head_dim, len(grids[0]))
    ang = inv.view(1, -1, 1) * ids.to(torch.float64).unsqueeze(-2)  # [L, F, axes]
    ang = ang.reshape(ids.shape[0], -1)
    fc = torch.polar(torch.ones(1, dtype=torch.float64), ang)   # rope.py:54 (bit-equal cos/sin)
    return fc.real.contiguous(), fc.imag.contiguous()


def apply_rotary(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """Rotate the first `cos.shape[-1]` (re,im) pairs of each head, fp32 math - rope.py:19-27.

    x [L, H, D]; cos/sin [L, R]; pairs are (x[2j], x[2j+1]); trailing D/2-R pairs untouched.
    """
    xf = x.float()
    L, H, D = xf.shape
    pairs = xf.reshape(L, H, D // 2, 2)
    R = cos.shape[-1]
    c = cos.to(torch.float32).unsqueeze(1)
    s = sin.to(torch.float32).unsqueeze(1)
    re, im = pairs[..., :R, 0], pairs[..., :R, 1]
    out = pairs.clone()
    out[..., :R, 0] = re * c - im * s
    out[..., :R, 1] = re * s + im * c
    return out.reshape(L, H, D).to(x.dtype)


# ----------------------------------------------------------------------------------------------
# third-party ops the reference calls (definitions, not reference code)
# ----------------------------------------------------------------------------------------------
def rmsnorm(x: Tensor, weight: Tensor, eps: float = RMS_EPS) -> Tensor:
    """flash_attn RMSNorm: fp32 `x * rsqrt(mean(x^2)+eps) * w`, output in x dtype."""
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * weight.float()
    return y.to(x.dtype)


def attention_varlen(q: Tensor, k: Tensor, v: Tensor, cu_seqlens: Sequence[int]) -> Tensor:
    """flash_attn_varlen_func semantics at transformer.py:100: per-sequence, non-causal
    softmax(q k^T / sqrt(D)) v, GQA (q head h uses kv head h // (Hq/Hkv)).

    fp32 inputs: fp32 math throughout.  16-bit inputs (flash-attn only takes fp16 / bf16): the published
    FlashAttention-2 arithmetic - scores and softmax statistics in fp32, the unnormalised probabilities exp(s - max) rounded
    to the input dtype as the operand of the second product, fp32 accumulation, one division by the fp32 row sum at the end."""
    L, Hq, D = q.shape
    Hkv = k.shape[1]
    rep = Hq // Hkv
    out = torch.empty_like(q, dtype=torch.float32)
    scale = D ** -0.5
    low = q.dtype in (torch.bfloat16, torch.float16)
    for b in range(len(cu_seqlens) - 1):
        s, e = int(cu_seqlens[b]), int(cu_seqlens[b + 1])
        qb = q[s:e].float().transpose(0, 1)                       # [Hq, S, D]
        kb = k[s:e].float().transpose(0, 1).repeat_interleave(rep, dim=0)
        vb = v[s:e].float().transpose(0, 1).repeat_interleave(rep, dim=0)
        sc = qb @ kb.transpose(1, 2) * scale
        if low:
            pt = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
            o = (pt.to(q.dtype).float() @ vb) / pt.sum(dim=-1, keepdim=True)
        else:
            o = torch.softmax(sc, dim=-1) @ vb
        out[s:e] = o.transpose(0, 1)
    return out.to(q.dtype)


# ----------------------------------------------------------------------------------------------
# model/base/transformer.py
# ----------------------------------------------------------------------------------------------
def attn_sublayer(x: Tensor, sd: Dict[str, Tensor], prefix: str, heads: Tuple[int, int],
                  cos: Tensor, sin: Tensor, cu_seqlens: Sequence[int]) -> Tensor:
    """Attn.forward - transformer.py:85-104 (split order q, gate, k, v at line 87)."""
    hq, hkv = heads
    d = x.shape[-1]
    hd = d // hq
    g = hd * hkv
    h = rmsnorm(x, sd[prefix + "pre_ln.weight"])
    qkv = F.linear(h, sd[prefix + "to_qkv.weight"].to(h.dtype))
    q, gate, k, v = qkv.split([d, d, g, g], dim=-1)
    q = apply_rotary(q.unflatten(-1, (hq, hd)), cos, sin)
    k = apply_rotary(k.unflatten(-1, (hkv, hd)), cos, sin)
    v = v.unflatten(-1, (hkv, hd))
    o = attention_varlen(q, k, v, cu_seqlens).flatten(-2)
    o = o * torch.sigmoid(gate)
    return F.linear(o, sd[prefix + "out_proj.weight"].to(o.dtype))


def geglu_sublayer(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """GEGLU.forward - transformer.py:47-56 (`x, gate = chunk(2)`; exact-erf gelu on gate)."""
    h = rmsnorm(x, sd[prefix + "norm.weight"])
    h = F.linear(h, sd[prefix + "w12.weight"].to(h.dtype))
    a, gate = h.chunk(2, dim=-1)
    h = F.gelu(gate) * a
    return F.linear(h, sd[prefix + "w3.weight"].to(h.dtype))


def transformer_stack(x: Tensor, sd: Dict[str, Tensor], prefix: str, layers: int,
                      heads: Tuple[int, int], cos: Tensor, sin: Tensor,
                      cu_seqlens: Sequence[int]) -> Tensor:
    """ResidualAttentionBlock.forward - transformer.py:126-146.

    Layer 0 is a pre-LN residual; layers >= 1 are KEEL: x = post_ln(alpha*x + f(x)), alpha = 2*layers.
    """
    alpha = 2 * layers
    for i in range(layers):
        ap = f"{prefix}attn_layer.{i}."
        fp = f"{prefix}ffd_layer.{i}."
        if i == 0:
            x = x + attn_sublayer(x, sd, ap, heads, cos, sin, cu_seqlens)
            x = x + geglu_sublayer(x, sd, fp)
        else:
            x = alpha * x + attn_sublayer(x, sd, ap, heads, cos, sin, cu_seqlens)
            x = rmsnorm(x, sd[f"{prefix}attn_post_ln.{i - 1}.weight"])
            x = alpha * x + geglu_sublayer(x, sd, fp)
            x = rmsnorm(x, sd[f"{prefix}ffd_post_ln.{i - 1}.weight"])
    return x


# ----------------------------------------------------------------------------------------------
# model/base/blocks.py
# ----------------------------------------------------------------------------------------------
def batch_metadata(pixel_grids: Sequence[Sequence[int]], token_counts: Sequence[int],
                   patch: Sequence[int]):
    """grids//patch, grid_sizes, cu_seqlens, latent-first bool mask - blocks.py:80-86 / 154-160."""
    grids = [[int(g) // int(p) for g, p in zip(pg, patch)] for pg in pixel_grids]
    sizes = [math.prod(g) for g in grids]
    counts = [int(k) for k in token_counts]
    cu = [0]
    mask: List[bool] = []
    for k, p in zip(counts, sizes):
        cu.append(cu[-1] + k + p)
        mask += [True] * k + [False] * p
    return grids, sizes, counts, cu, torch.tensor(mask, dtype=torch.bool)


def encoder_forward(videos: Sequence[Tensor], token_counts: Sequence[int], sd: Dict[str, Tensor],
                    model_size: str = "tiny", patch: Sequence[int] = (4, 8, 8),
                    prefix: str = "") -> Tensor:
    """TiTokEncoder.forward - blocks.py:71-104.  Returns [sum(K), out_channels]."""
    width, layers, heads = model_dims(model_size)
    dtype = videos[0].dtype
    pix = [v.shape[1:] for v in videos]
    grids, sizes, counts, cu, mask = batch_metadata(pix, token_counts, patch)
    cos, sin = rope_table(grids, counts, width // heads[0])

    patches = torch.cat([patchify(v, patch) for v in videos], dim=0)
    patches = F.linear(patches, sd[prefix + "proj_in.weight"].to(dtype), sd[prefix + "proj_in.bias"].to(dtype))
    mt = sd[prefix + "mask_token"].to(dtype)
    x = torch.zeros(mask.shape[0], width, dtype=dtype)
    x[mask] = rmsnorm(mt.expand(-1, width), sd[prefix + "ln_pre_t.weight"])
    x[~mask] = rmsnorm(patches + mt, sd[prefix + "ln_pre_p.weight"])

    x = transformer_stack(x, sd, prefix + "model_layers.", layers, heads, cos, sin, cu)

    tokens = rmsnorm(x[mask], sd[prefix + "ln_post.weight"])
    return F.linear(tokens, sd[prefix + "proj_out.weight"].to(dtype), sd[prefix + "proj_out.bias"].to(dtype))


def decoder_forward(tokens: Tensor, token_counts: Sequence[int], pixel_grids: Sequence[Sequence[int]],
                    sd: Dict[str, Tensor], model_size: str = "tiny",
                    patch: Sequence[int] = (4, 8, 8), out_channels: int = 3,
                    prefix: str = "") -> List[Tensor]:
    """TiTokDecoder.forward - blocks.py:148-177.  Returns list of [C,T,H,W]."""
    width, layers, heads = model_dims(model_size)
    dtype = tokens.dtype
    grids, sizes, counts, cu, mask = batch_metadata(pixel_grids, token_counts, patch)
    cos, sin = rope_table(grids, counts, width // heads[0])

    mt = sd[prefix + "mask_token"].to(dtype)
    x = torch.zeros(mask.shape[0], width, dtype=dtype)
    h = F.linear(tokens, sd[prefix + "proj_in.weight"].to(dtype), sd[prefix + "proj_in.bias"].to(dtype))
    x[mask] = rmsnorm(h + mt, sd[prefix + "ln_pre_t.weight"])
    x[~mask] = rmsnorm(mt.expand(-1, width), sd[prefix + "ln_pre_p.weight"])

    x = transformer_stack(x, sd, prefix + "model_layers.", layers, heads, cos, sin, cu)

    p = rmsnorm(x[~mask], sd[prefix + "ln_post.weight"])
    p = F.linear(p, sd[prefix + "proj_out.weight"].to(dtype), sd[prefix + "proj_out.bias"].to(dtype))
    outs = []
    for chunk, g in zip(torch.split(p, sizes, dim=0), grids):
        outs.append(unpatchify(chunk, g, patch, out_channels))
    return outs


# ----------------------------------------------------------------------------------------------
# model/quantizer/fsq.py
# ----------------------------------------------------------------------------------------------
def fsq_tables(levels: Sequence[int]):
    lv = torch.tensor(list(levels), dtype=torch.int32)
    basis = torch.cumprod(torch.tensor([1] + list(levels[:-1])), dim=0, dtype=torch.int32)  # fsq.py:66
    return lv, basis


def fsq_bound(z: Tensor, levels: Sequence[int], eps: float = 1e-3) -> Tensor:
    """fsq.py:78-83 (fp32)."""
    lv, _ = fsq_tables(levels)
    half_l = (lv - 1) * (1 + eps) / 2
    offset = torch.where(lv % 2 == 0, 0.5, 0.0)
    shift = (offset / half_l).atanh()
    return (z + shift).tanh() * half_l - offset


def fsq_forward(z: Tensor, levels: Sequence[int]):
    """FSQ.forward - fsq.py:123-135: fp32 bound -> round(half-even) -> /half_width -> index.

    Returns (codes in z dtype, int32 indices, fp32 bounded values before rounding).
    """
    lv, basis = fsq_tables(levels)
    zf = z.float()
    bounded = fsq_bound(zf, levels)
    q = bounded + (bounded.round() - bounded).detach()   # round_ste (fsq.py:48-51): round forward, identity backward
    half_width = lv // 2
    codes = q / half_width                   # fsq.py:85-90
    zhat = codes * half_width + half_width   # fsq.py:92-94
    idx = (zhat * basis).sum(dim=-1).to(torch.int32)  # fsq.py:105-109
    return codes.to(z.dtype), idx, bounded


def fsq_indices_to_codes(indices: Tensor, levels: Sequence[int]) -> Tensor:
    """fsq.py:100-121: (idx // basis) % levels, then (lvl - hw) / hw."""
    lv, basis = fsq_tables(levels)
    lvl = (indices.unsqueeze(-1) // basis) % lv
    hw = lv // 2
    return (lvl - hw) / hw


def fsq_margin(bounded: Tensor) -> Tensor:
    """Per-token rounding margin min_c(0.5 - |b - round(b)|)."""
    return (0.5 - (bounded - bounded.round()).abs()).min(dim=-1).values


# ----------------------------------------------------------------------------------------------
# model/titok.py
# ----------------------------------------------------------------------------------------------
def titok_encode(videos, token_counts, sd, levels, enc_size="tiny", patch=(4, 8, 8)):
    """TiTok.encode - titok.py:47-52.  Returns (codes, indices, z, bounded)."""
    z = encoder_forward(videos, token_counts, sd, enc_size, patch, prefix="encoder.")
    codes, idx, bounded = fsq_forward(z, levels)
    return codes, idx, z, bounded


def titok_decode(codes, token_counts, pixel_grids, sd, dec_size="tiny", patch=(4, 8, 8)):
    """TiTok.decode - titok.py:64-66."""
    return decoder_forward(codes, token_counts, pixel_grids, sd, dec_size, patch, 3, prefix="decoder.")


def titok_decode_indices(indices, pixel_grids, token_counts, sd, levels, dec_size="tiny",
                         patch=(4, 8, 8), dtype=torch.float32):
    """TiTok.decode_indices - titok.py:54-62."""
    codes = fsq_indices_to_codes(indices, levels).to(dtype)
    return titok_decode(codes, token_counts, pixel_grids, sd, dec_size, patch)


def titok_forward(videos, token_counts, sd, levels, enc_size="tiny", dec_size="tiny", patch=(4, 8, 8)):
    """TiTok.forward - titok.py:68-74.  Returns (recon list, indices, z, bounded)."""
    pix = [v.shape[1:] for v in videos]
    codes, idx, z, bounded = titok_encode(videos, token_counts, sd, levels, enc_size, patch)
    recon = titok_decode(codes, token_counts, pix, sd, dec_size, patch)
    return recon, idx, z, bounded


# ----------------------------------------------------------------------------------------------
# train_utils/codebook_logging.py
# ----------------------------------------------------------------------------------------------
def codebook_scores(samples: Sequence[Tensor], codebook_size: int):
    """CodebookLogger.get_scores - codebook_logging.py:19-32: sum of per-sample bincounts ->
    usage percent and entropy (nats) of the normalised histogram."""
    freq = torch.zeros(codebook_size)
    for s in samples:
        freq += torch.bincount(s.to(torch.int64), minlength=codebook_size)
    usage = float((freq.count_nonzero() / codebook_size) * 100)
    p = freq.double() / freq.double().sum()
    nz = p[p > 0]
    ent = float(-(nz * nz.log()).sum())
    return usage, ent, freq
