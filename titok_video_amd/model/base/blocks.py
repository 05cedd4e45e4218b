"""TiTokEncoder / TiTokDecoder: the reference's tower API (model/base/blocks.py:31-104, 108-177) on the HIP path.

Same constructor arguments, `forward` signatures and state-dict keys as the reference, so checkpoints and the
reference's train.py / loss_module.py see the same modules.  The modules only HOLD parameters; all arithmetic runs
in libtitok_hip.so (csrc/): `forward` builds the host-side batch plan, packs weights into the compute dtype and
makes ONE C-ABI call per tower (ttv_encoder_forward / ttv_decoder_forward), which enqueues every kernel on the
current torch stream without synchronising.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from ... import _lib
from ...plan import BatchPlan, get_plan, host_ints
from .utils import LinearWeight, RMSNormWeight, geglu_inner_dim, get_model_dims


class _AttnParams(nn.Module):
    """Parameters of reference `Attn` (transformer.py:69-83): pre_ln, to_qkv [(2d+2g), d], out_proj [d, d]."""

    def __init__(self, dim: int, heads: Sequence[int]):
        super().__init__()
        q_heads, kv_heads = heads
        gqa = (dim // q_heads) * kv_heads
        self.pre_ln = RMSNormWeight(dim)
        self.to_qkv = LinearWeight(dim, 2 * gqa + 2 * dim, bias=False)
        self.out_proj = LinearWeight(dim, dim, bias=False)


class _GegluParams(nn.Module):
    """Parameters of reference `GEGLU` (transformer.py:36-45): norm, w12 [2I, d], w3 [d, I]."""

    def __init__(self, dim: int, mult: float = 4.0):
        super().__init__()
        inner = geglu_inner_dim(dim, mult)
        self.norm = RMSNormWeight(dim)
        self.w12 = LinearWeight(dim, 2 * inner, bias=False)
        self.w3 = LinearWeight(inner, dim, bias=False)


class ResidualAttentionBlock(nn.Module):
    """Parameter container with the key layout of reference `ResidualAttentionBlock` (transformer.py:107-123)."""

    def __init__(self, embed_dim=512, heads=(8, 2), mlp_ratio=4, num_layer=2):
        super().__init__()
        self.num_layer = num_layer
        self.alpha = num_layer * 2
        self.attn_layer = nn.ModuleList([_AttnParams(embed_dim, heads) for _ in range(num_layer)])
        self.ffd_layer = nn.ModuleList([_GegluParams(embed_dim, mlp_ratio) for _ in range(num_layer)])
        self.attn_post_ln = nn.ModuleList([RMSNormWeight(embed_dim) for _ in range(num_layer - 1)])
        self.ffd_post_ln = nn.ModuleList([RMSNormWeight(embed_dim) for _ in range(num_layer - 1)])


def _versions(params) -> tuple:
    return tuple((p.data_ptr(), p._version, p.dtype) for p in params)


class _Tower(nn.Module):
    kind = _lib.TTV_ENCODER

    def _setup(self, model_size, patch_size, pix_channels, token_size):
        self.model_size = model_size
        self.patch = tuple(int(p) for p in patch_size)
        self.patch_size = torch.tensor(self.patch, dtype=torch.int32)
        self.width, self.num_layers, self.heads, mlp_ratio = get_model_dims(model_size)
        self.inner = geglu_inner_dim(self.width, mlp_ratio)
        self.pix_channels = pix_channels
        self.token_size = token_size
        self._pack = None
        self._pack_key = None
        self._ws = {}
        return mlp_ratio

    # ---- weight packing -------------------------------------------------------------------------
    def _patch_perm(self) -> torch.Tensor:
        """Index map from the kernels' (c,pt,ph,pw) patch-vector order to the reference's (pt,ph,pw,c) order
        (utils.py:32): packed[..., j] = reference[..., perm[j]]."""
        pt, ph, pw = self.patch
        c = self.pix_channels
        idx = torch.arange(pt * ph * pw * c).reshape(pt, ph, pw, c).permute(3, 0, 1, 2).reshape(-1)
        return idx

    def _packed(self, dtype: torch.dtype, device) -> "_WeightPack":
        params = list(self.parameters())
        key = (dtype, str(device), _versions(params))
        if self._pack is None or self._pack_key != key:
            self._pack = _WeightPack(self, dtype, device)
            self._pack_key = key
        return self._pack

    def _dims(self, dtype_code: int) -> _lib.TowerDims:
        return _lib.TowerDims(kind=self.kind, dtype=dtype_code, width=self.width, layers=self.num_layers,
                              q_heads=self.heads[0], kv_heads=self.heads[1], head_dim=self.width // self.heads[0],
                              inner=self.inner, patch_t=self.patch[0], patch_h=self.patch[1], patch_w=self.patch[2],
                              pix_channels=self.pix_channels, token_size=self.token_size, eps=1e-5,
                              alpha=float(self.model_layers.alpha))

    def _workspace(self, dims: _lib.TowerDims, plan: BatchPlan, device) -> torch.Tensor:
        need = _lib.lib().ttv_tower_workspace_bytes(C.byref(dims), C.byref(plan.batch_for(self.heads[0], self.heads[1])))
        if need < 0:
            _lib.check(1, "ttv_tower_workspace_bytes")
        key = (str(device), dims.dtype)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(int(need), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def _plan(self, pixel_grids, token_counts, device) -> BatchPlan:
        return get_plan(pixel_grids, token_counts, self.patch, device)


class _WeightPack:
    """Compute-dtype copies of a tower's linear weights (what autocast does per call in the reference), the
    patch-order permutation folded into proj_in / proj_out, fp32 norm gains, and the C structs pointing at them."""

    def __init__(self, tower: _Tower, dtype: torch.dtype, device):
        keep: List[torch.Tensor] = []

        def lin(p):
            t = p.detach().to(device=device, dtype=dtype).contiguous()
            keep.append(t)
            return t.data_ptr()

        def gain(p):
            t = p.detach().to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        fold = dtype == torch.bfloat16 and tower.width == 256

        def folded(w, g):
            """w * gain[None, :] in fp32, then the compute dtype: lets the K=256 GEMM kernel absorb the pre-norm."""
            if not fold:
                return None
            t = (w.detach().to(device=device, dtype=torch.float32) * g.detach().to(device=device, dtype=torch.float32)[None, :])
            t = t.to(dtype).contiguous()
            keep.append(t)
            return t.data_ptr()

        def w3_permuted(w):
            """Columns of w3 reordered inside every 32-column panel to the k order of the fused feed-forward kernel
            (csrc/ttv_mlp.hip): new[:, 32p + 8q + j] = old[:, 32p + (j < 4 ? 4q + j : 16 + 4q + (j - 4))]."""
            if not fold or w.shape[1] % 32:
                return None
            q = torch.arange(4).view(4, 1)
            j = torch.arange(8).view(1, 8)
            within = torch.where(j < 4, 4 * q + j, 16 + 4 * q + (j - 4)).reshape(-1)            # [32]
            cols = (torch.arange(w.shape[1] // 32).view(-1, 1) * 32 + within.view(1, -1)).reshape(-1).to(device)
            t = w.detach().to(device=device, dtype=dtype)[:, cols].contiguous()
            keep.append(t)
            return t.data_ptr()

        perm = tower._patch_perm().to(device)
        if tower.kind == _lib.TTV_ENCODER:
            w_in = tower.proj_in.weight.detach().to(device)[:, perm]
            b_out, w_out = tower.proj_out.bias, tower.proj_out.weight
        else:
            w_in = tower.proj_in.weight
            w_out = tower.proj_out.weight.detach().to(device)[perm, :]
            b_out = tower.proj_out.bias.detach().to(device)[perm]
        ml = tower.model_layers
        n = tower.num_layers
        self.layers = (_lib.LayerWeights * n)()
        for i in range(n):
            a, f = ml.attn_layer[i], ml.ffd_layer[i]
            self.layers[i] = _lib.LayerWeights(
                pre_ln=gain(a.pre_ln.weight), to_qkv=lin(a.to_qkv.weight), out_proj=lin(a.out_proj.weight),
                ffd_norm=gain(f.norm.weight), w12=lin(f.w12.weight), w3=lin(f.w3.weight),
                attn_post_ln=gain(ml.attn_post_ln[i - 1].weight) if i > 0 else None,
                ffd_post_ln=gain(ml.ffd_post_ln[i - 1].weight) if i > 0 else None,
                to_qkv_pn=folded(a.to_qkv.weight, a.pre_ln.weight), w12_pn=folded(f.w12.weight, f.norm.weight),
                w3_perm=w3_permuted(f.w3.weight))
        self.struct = _lib.TowerWeights(
            proj_in_w=lin(w_in), proj_in_b=lin(tower.proj_in.bias), mask_token=gain(tower.mask_token),
            ln_pre_t=gain(tower.ln_pre_t.weight), ln_pre_p=gain(tower.ln_pre_p.weight), ln_post=gain(tower.ln_post.weight),
            proj_out_w=lin(w_out), proj_out_b=lin(b_out), layers=self.layers)
        self.keep = keep


class TiTokEncoder(_Tower):
    """Reference TiTokEncoder (blocks.py:31-104): list of [C,T,H,W] clips -> latent tokens [sum(K), out_channels]."""
    kind = _lib.TTV_ENCODER

    def __init__(self, model_size="tiny", patch_size=(4, 8, 8), in_channels=3, out_channels=5):
        super().__init__()
        mlp_ratio = self._setup(model_size, patch_size, in_channels, out_channels)
        self.in_channels = in_channels
        scale = self.width ** -0.5
        self.proj_in = LinearWeight(in_channels * math.prod(self.patch), self.width)
        self.mask_token = nn.Parameter(scale * torch.randn(1, 1))
        self.ln_pre_t = RMSNormWeight(self.width)
        self.ln_pre_p = RMSNormWeight(self.width)
        self.model_layers = ResidualAttentionBlock(self.width, self.heads, mlp_ratio, self.num_layers)
        self.ln_post = RMSNormWeight(self.width)
        self.proj_out = LinearWeight(self.width, self.token_size, bias=True)

    def run(self, videos: Sequence[torch.Tensor], token_counts, grids=None, fsq_params=None, want_z=True,
            want_bounded=False):
        """Fused tower + FSQ tail.  Returns dict(z fp32 | None, codes, indices, bounded) (codes/indices None without fsq)."""
        v0 = videos[0]
        _lib.require_gpu(v0, "TiTokEncoder")
        device, dtype = v0.device, v0.dtype
        code = _lib.dtype_code(dtype)
        counts = host_ints(token_counts)
        pix = [tuple(v.shape[1:]) for v in videos] if grids is None else host_ints(grids)
        for v, g in zip(videos, pix):
            if v.dtype != dtype or v.device != device or tuple(v.shape[1:]) != tuple(g) or v.shape[0] != self.pix_channels:
                raise ValueError("clips must share dtype/device, be [C,T,H,W] and match `grids`")
        plan = self._plan(pix, counts, device)
        dims = self._dims(code)
        pack = self._packed(dtype, device)
        ws = self._workspace(dims, plan, device)
        clips = [v if v.is_contiguous() else v.contiguous() for v in videos]
        n, c = plan.sum_tokens, self.token_size
        z = torch.empty((n, c), dtype=torch.float32, device=device) if (want_z or fsq_params is None) else None
        codes = indices = bounded = None
        if fsq_params is not None:
            codes = torch.empty((n, c), dtype=dtype, device=device)
            indices = torch.empty((n,), dtype=torch.int32, device=device)
            if want_bounded:
                bounded = torch.empty((n, c), dtype=torch.float32, device=device)
        batch = plan.batch_for(self.heads[0], self.heads[1])
        rc = _lib.lib().ttv_encoder_forward(
            C.byref(dims), C.byref(pack.struct), C.byref(batch), _lib.ptr_array(clips),
            C.byref(fsq_params) if fsq_params is not None else None, _lib.ptr(z), _lib.ptr(codes), _lib.ptr(indices),
            _lib.ptr(bounded), ws.data_ptr(), ws.numel(), _lib.stream_ptr(device))
        _lib.check(rc, "ttv_encoder_forward")
        return {"z": z, "codes": codes, "indices": indices, "bounded": bounded}

    def forward(self, videos, token_counts, grids=None):
        out = self.run(videos, token_counts, grids, None, want_z=True)
        return out["z"].to(videos[0].dtype)


class TiTokDecoder(_Tower):
    """Reference TiTokDecoder (blocks.py:108-177): tokens [sum(K), in_channels] -> list of [C,T,H,W] clips."""
    kind = _lib.TTV_DECODER

    def __init__(self, model_size="tiny", patch_size=(4, 8, 8), in_channels=5, out_channels=3):
        super().__init__()
        mlp_ratio = self._setup(model_size, patch_size, out_channels, in_channels)
        self.out_channels = out_channels
        scale = self.width ** -0.5
        self.proj_in = LinearWeight(self.token_size, self.width, bias=True)
        self.mask_token = nn.Parameter(scale * torch.randn(1, 1))
        self.ln_pre_t = RMSNormWeight(self.width)
        self.ln_pre_p = RMSNormWeight(self.width)
        self.model_layers = ResidualAttentionBlock(self.width, self.heads, mlp_ratio, self.num_layers)
        self.ln_post = RMSNormWeight(self.width)
        self.proj_out = LinearWeight(self.width, out_channels * math.prod(self.patch))

    def forward(self, tokens: torch.Tensor, token_counts, grids):
        _lib.require_gpu(tokens, "TiTokDecoder")
        device, dtype = tokens.device, tokens.dtype
        code = _lib.dtype_code(dtype)
        counts = host_ints(token_counts)
        pix = [tuple(int(v) for v in g) for g in host_ints(grids)]
        plan = self._plan(pix, counts, device)
        if tokens.shape != (plan.sum_tokens, self.token_size):
            raise ValueError(f"tokens must be [{plan.sum_tokens}, {self.token_size}], got {tuple(tokens.shape)}")
        dims = self._dims(code)
        pack = self._packed(dtype, device)
        ws = self._workspace(dims, plan, device)
        tokens = tokens.contiguous()
        sizes = [self.out_channels * t * h * w for (t, h, w) in pix]
        flat = torch.empty(sum(sizes), dtype=dtype, device=device)
        outs, off = [], 0
        for (t, h, w), n in zip(pix, sizes):
            outs.append(flat[off:off + n].view(self.out_channels, t, h, w))
            off += n
        batch = plan.batch_for(self.heads[0], self.heads[1])
        rc = _lib.lib().ttv_decoder_forward(C.byref(dims), C.byref(pack.struct), C.byref(batch), tokens.data_ptr(),
                                            _lib.ptr_array(outs), ws.data_ptr(), ws.numel(), _lib.stream_ptr(device))
        _lib.check(rc, "ttv_decoder_forward")
        return outs
