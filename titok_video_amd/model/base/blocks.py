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
import os
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


_PACK_CHECK = os.environ.get("TTV_PACK_CHECK", "0") == "1"


class _Tower(nn.Module):
    kind = _lib.TTV_ENCODER

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .bfloat16() may replace parameter objects: drop the cached parameter list and gradient layout
        self.__dict__.pop("_param_list_cache", None)
        self.__dict__.pop("_grad_layout_cache", None)
        out = super()._apply(fn, *args, **kwargs)
        self.__dict__.pop("_param_list_cache", None)
        self.__dict__.pop("_grad_layout_cache", None)
        return out

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
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packs())
        return mlp_ratio

    # ---- weight packing -------------------------------------------------------------------------
    def _patch_perm(self) -> torch.Tensor:
        """Index map from the kernels' (c,pt,ph,pw) patch-vector order to the reference's (pt,ph,pw,c) order
        (utils.py:32): packed[..., j] = reference[..., perm[j]]."""
        pt, ph, pw = self.patch
        c = self.pix_channels
        idx = torch.arange(pt * ph * pw * c).reshape(pt, ph, pw, c).permute(3, 0, 1, 2).reshape(-1)
        return idx

    def _patch_perm_on(self, device) -> torch.Tensor:
        """The same index map resident on `device`.  Cached: a host->device copy of a pageable tensor synchronises with all
        queued GPU work, which would serialise the backward pass (and every per-step weight repack) with the host."""
        cache = self.__dict__.setdefault("_perm_cache", {})
        key = str(device)
        if key not in cache:
            cache[key] = self._patch_perm().to(device)
        return cache[key]

    def invalidate_packs(self) -> None:
        """Forget the packed weight copies.  The cache key is (storage pointer, tensor version, dtype) of every parameter, which
        catches optimizer steps, load_state_dict and in-place ops under no_grad - but NOT writes through `.data` (they do not
        bump the version counter: `p.data.mul_(2)`, `trunc_normal_(p.data)`, EMA by `p.data.copy_`).  Code that updates weights
        that way calls this afterwards; `init_weights` and `load_state_dict` do it themselves.  TTV_PACK_CHECK=1 (debugging)
        additionally compares a content checksum on every call."""
        self._retire_pack()
        self._pack_key = None

    def _retire_pack(self) -> None:
        """The old pack's memory goes back to the allocator of the stream that built it: every other stream that read it (forwards
        in flight in a ForwardPipeline) must be finished from that stream's point of view first."""
        old, self._pack = self._pack, None
        if old is not None and old.device.type == "cuda":
            cur = torch.cuda.current_stream(old.device)
            build = getattr(old, "build_stream", None)
            for st in old.reader_streams.values():
                if st != cur:
                    cur.wait_stream(st)
                # the freed buffers return to the BUILD stream's pool: when that is not the current stream (a version change noticed
                # inside a ForwardPipeline side stream) it must see the readers as finished too, or it could hand the memory out
                # again while another stream is still reading the old pack
                if build is not None and build != cur and st != build:
                    build.wait_stream(st)

    def _packed(self, dtype: torch.dtype, device) -> "_WeightPack":
        params = self._param_list()
        key = (dtype, str(device), _versions(params), str(getattr(self, "fp8_linears", False)), bool(getattr(self, "f32_split3", False)))
        if _PACK_CHECK:
            key = key + (float(sum(p.detach().double().sum() for p in params)),)
        if self._pack is None or self._pack_key != key:
            self._retire_pack()
            self._pack = _WeightPack(self, dtype, device)
            self._pack_key = key
        self._pack.use_on_current_stream()
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
        # one scratch buffer per (device, dtype, STREAM): forwards issued on different streams may run concurrently
        # (titok_video_amd.pipeline.ForwardPipeline) and must not share it
        stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
        key = (str(device), dims.dtype, stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(int(need), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def _plan(self, pixel_grids, token_counts, device) -> BatchPlan:
        return get_plan(pixel_grids, token_counts, self.patch, device)

    # ---- training (autograd) support -------------------------------------------------------------
    def _wants_grad(self, *inputs) -> bool:
        if not torch.is_grad_enabled():
            return False
        return any(p.requires_grad for p in self._param_list()) or any(getattr(t, "requires_grad", False) for t in inputs)

    def _param_list(self):
        """Parameters in registration order, cached (module.parameters() walks the module tree on every call)."""
        pl = self.__dict__.get("_param_list_cache")
        if pl is None:
            pl = list(self.parameters())
            self.__dict__["_param_list_cache"] = pl
        return pl

    def _grad_layout(self):
        """(parameters, packed shapes, sizes, offsets, total) of the flat fp32 gradient buffer, in the order of the C struct; cached."""
        lay = self.__dict__.get("_grad_layout_cache")
        if lay is None:
            ml = self.model_layers
            named = []   # (parameter, packed shape)
            named.append((self.proj_in.weight, tuple(self.proj_in.weight.shape)))
            named.append((self.proj_in.bias, tuple(self.proj_in.bias.shape)))
            named.append((self.mask_token, (1,)))
            for m in (self.ln_pre_t, self.ln_pre_p, self.ln_post):
                named.append((m.weight, (self.width,)))
            named.append((self.proj_out.weight, tuple(self.proj_out.weight.shape)))
            named.append((self.proj_out.bias, tuple(self.proj_out.bias.shape)))
            for i in range(self.num_layers):
                a, f = ml.attn_layer[i], ml.ffd_layer[i]
                for prm in (a.pre_ln.weight, a.to_qkv.weight, a.out_proj.weight, f.norm.weight, f.w12.weight, f.w3.weight):
                    named.append((prm, tuple(prm.shape)))
                if i > 0:
                    named.append((ml.attn_post_ln[i - 1].weight, (self.width,)))
                    named.append((ml.ffd_post_ln[i - 1].weight, (self.width,)))
            sizes = [int(math.prod(sh)) for _, sh in named]
            offs, tot = [], 0
            for n in sizes:
                offs.append(tot)
                tot += (n + 63) // 64 * 64
            lay = ([p for p, _ in named], [sh for _, sh in named], sizes, offs, tot)
            self.__dict__["_grad_layout_cache"] = lay
        return lay

    def _grad_buffers(self, device):
        """Zeroed flat fp32 gradient buffer in the PACKED weight layout + the C struct pointing into it."""
        params, shapes, sizes, offs, tot = self._grad_layout()
        flat = torch.zeros(tot, dtype=torch.float32, device=device)
        base = flat.data_ptr()
        ptr = [base + 4 * o for o in offs]
        lay = (_lib.LayerGrads * self.num_layers)()
        k = 8
        for i in range(self.num_layers):
            L = lay[i]
            L.pre_ln, L.to_qkv, L.out_proj, L.ffd_norm, L.w12, L.w3 = ptr[k:k + 6]
            k += 6
            if i > 0:
                L.attn_post_ln, L.ffd_post_ln = ptr[k], ptr[k + 1]
                k += 2
        st = _lib.TowerGrads(proj_in_w=ptr[0], proj_in_b=ptr[1], mask_token=ptr[2], ln_pre_t=ptr[3], ln_pre_p=ptr[4], ln_post=ptr[5],
                             proj_out_w=ptr[6], proj_out_b=ptr[7], layers=lay)
        # data-parallel training (dp.GradReducer attached): one event per layer, recorded by the library when the layer's slice is final
        events = None
        if self.__dict__.get("_grad_reducer") is not None and flat.is_cuda:
            events = [torch.cuda.Event() for _ in range(self.num_layers)]
            cur = torch.cuda.current_stream(flat.device)
            for e in events:
                e.record(cur)                      # creates the underlying hipEvent (lazy in torch) so that its handle can be passed
            arr = (_lib.vp * self.num_layers)(*[e.cuda_event for e in events])
            st.layer_done_events = arr
            lay = (lay, arr)                       # keep the ctypes arrays alive with the struct
        self.__dict__["_last_grad_events"] = events
        return flat, st, lay

    def _reduce_and_finish(self, flat: torch.Tensor):
        """After the backward kernels are enqueued: hand the flat gradient buffer to the attached dp.GradReducer slice by slice
        (layer slices in the order the backward completes them, then the head / tail parameters, which are final only at the end of
        the backward), and build the per-parameter gradients behind the reductions."""
        red = self.__dict__.get("_grad_reducer")
        events = self.__dict__.pop("_last_grad_events", None)
        if red is None or events is None:
            return self._finish_grads(flat)
        _params, _shapes, sizes, offs, tot = self._grad_layout()
        bounds = []                                    # [lo, hi) of every layer in the flat buffer (entries 8.. of the layout)
        k = 8
        for i in range(self.num_layers):
            n_i = 6 if i == 0 else 8
            lo = offs[k]
            hi = offs[k + n_i] if k + n_i < len(offs) else tot
            bounds.append((lo, hi))
            k += n_i
        for i in reversed(range(self.num_layers)):
            red.reduce_slice(flat, bounds[i][0], bounds[i][1], events[i])
        tail = torch.cuda.Event()
        tail.record(torch.cuda.current_stream(flat.device))
        red.reduce_slice(flat, 0, offs[8], tail)       # proj_in / mask_token / ln_pre / ln_post / proj_out
        with red.stream():
            out = self._finish_grads(flat)
        params = self._grad_layout()[0]
        red.reduced_ids.update(id(p) for p in params)
        if red.on_comm_stream:
            # produced on the communication stream: never handed to autograd (its AccumulateGrad does not wait for that stream);
            # the reducer assigns them to p.grad in finish(), behind the stream join
            red.defer([(p, out[id(p)]) for p in params if p.requires_grad and id(p) in out])
            return {}
        return out

    def _finish_grads(self, flat: torch.Tensor):
        """Flat packed fp32 gradients -> {id(parameter): gradient in the reference layout and the parameter's dtype}: one cast of
        the whole buffer (instead of a tiny kernel per parameter), one slice per parameter, the patch-order permutation undone."""
        params, shapes, sizes, offs, _tot = self._grad_layout()
        dt = params[0].dtype
        conv = flat if dt == torch.float32 else flat.to(dt)
        perm = None
        out = {}
        for p, sh, n, o in zip(params, shapes, sizes, offs):
            g = conv[o:o + n].view(sh)
            if p is self.mask_token:
                g = g.view(1, 1)
            elif (self.kind == _lib.TTV_ENCODER and p is self.proj_in.weight) or \
                 (self.kind == _lib.TTV_DECODER and (p is self.proj_out.weight or p is self.proj_out.bias)):
                if perm is None:
                    perm = self._patch_perm_on(flat.device)
                r = torch.empty_like(g)
                if p is self.proj_in.weight:
                    r[:, perm] = g
                else:
                    r[perm] = g          # rows of proj_out.weight / entries of proj_out.bias
                g = r
            out[id(p)] = g
        return out


class _WeightPack:
    """Compute-dtype copies of a tower's linear weights (what autocast does per call in the reference), the
    patch-order permutation folded into proj_in / proj_out, fp32 norm gains, and the C structs pointing at them."""

    def __init__(self, tower: _Tower, dtype: torch.dtype, device):
        keep: List[torch.Tensor] = []
        self.dtype, self.device = dtype, torch.device(device)
        self.lin_tensors: List[torch.Tensor] = []     # compute-dtype linear weights in creation order (see transposed())
        self.ready = None                              # event recorded on the building stream once every copy / pack kernel is queued
        self.reader_streams = {}                       # stream id -> stream, of every stream a forward read this pack on

        # fp32 towers in split-bf16 mode (tower.f32_split3, inference): the GEMM weights are handed over as split images
        # (ttv_split3_pack: per four k values hi0..3 | lo0..3 in bf16), same bytes and leading dimension as the fp32 matrix
        self.f32_split3 = bool(getattr(tower, "f32_split3", False)) and dtype == torch.float32

        def lin(p, gemm_weight=False):
            t = p.detach().to(device=device, dtype=dtype).contiguous()
            keep.append(t)
            self.lin_tensors.append(t)
            if gemm_weight and self.f32_split3 and t.dim() == 2 and t.shape[1] % 4 == 0:
                img = torch.empty_like(t)
                _lib.check(_lib.lib().ttv_split3_pack(t.data_ptr(), t.shape[1], img.data_ptr(), t.shape[1], t.shape[0], t.shape[1], _lib.stream_ptr(device)),
                           "ttv_split3_pack")
                keep.append(img)
                return img.data_ptr()
            return t.data_ptr()

        def gain(p):
            t = p.detach().to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        fold = dtype == torch.bfloat16 and tower.width == 256
        # other widths: the pre-norm gains are folded into to_qkv / w12 as well; the generic GEMM then scales its output rows by the
        # row statistic the producing kernel wrote (ttv_layer_weights.to_qkv_pn / w12_pn; TTV_FOLD_NORMS=0 keeps the stand-alone norms)
        fold_gen = dtype == torch.bfloat16 and tower.width != 256 and os.environ.get("TTV_FOLD_NORMS", "1") != "0"

        # q rows of the folded QKV weight carry head_dim^-0.5 * log2(e): the projection then emits the softmax exponent and the
        # attention kernel saves a multiply-add per score (TTV_ATTN_QSCALED).  Applied in fp32 before the one rounding to bf16.
        use_qs = dtype == torch.bfloat16 and os.environ.get("TTV_ATTN_QSCALE", "1") != "0"
        q_scale = 0.125 * 1.4426950408889634 if use_qs else None
        self.q_prescaled = 1 if ((fold or fold_gen) and use_qs) else 0

        def lin_qs(w):
            """Inference copy of to_qkv with the factor on its q rows, for towers whose QKV GEMM does not take the folded weight."""
            if fold or fold_gen or not use_qs:
                return None
            t = w.detach().to(device=device, dtype=torch.float32).clone()
            t[:tower.width] *= q_scale
            t = t.to(dtype).contiguous()
            keep.append(t)
            return t.data_ptr()

        f8_mode = getattr(tower, "fp8_linears", False)
        use_f8 = bool(f8_mode) and dtype == torch.bfloat16 and not fold and tower.width % 128 == 0
        # fp8_linears = "mx": all four linears of a layer in block-scaled (MX) e4m3, the pre-norm gains folded into the images
        # (ttv_layer_weights.to_qkv_mx ...); True: round 2's row-scaled to_qkv / w12
        use_mx = use_f8 and str(f8_mode).lower() == "mx" and tower.inner % 128 == 0
        if use_mx and use_qs:
            self.q_prescaled = 1          # the MX image of to_qkv carries the factor on its q rows

        def f8_mx(w, g=None, q_rows=0):
            """(e4m3 image, fp32 row factors, E8M0 block scales) of W' = w * gain[None, :] (q rows pre-scaled): ttv_quant_mx_fp8."""
            if not use_mx:
                return None, None, None
            t = w.detach().to(device=device, dtype=torch.float32)
            t = (t * g.detach().to(device=device, dtype=torch.float32)[None, :]) if g is not None else t.clone()
            if q_rows and q_scale is not None:
                t[:q_rows] *= q_scale
            t = t.contiguous()
            n_rows, k = t.shape
            q = torch.empty((n_rows, k), dtype=torch.uint8, device=device)
            sc = torch.empty(n_rows, dtype=torch.float32, device=device)
            mx = torch.zeros((n_rows, int(_lib.lib().ttv_mx_scale_bytes_per_row(k))), dtype=torch.uint8, device=device)
            _lib.check(_lib.lib().ttv_quant_mx_fp8(t.data_ptr(), _lib.TTV_F32, k, q.data_ptr(), k, mx.data_ptr(), sc.data_ptr(), n_rows, k,
                                                   _lib.stream_ptr(device)), "ttv_quant_mx_fp8")
            keep.extend([t, q, sc, mx])
            return q.data_ptr(), sc.data_ptr(), mx.data_ptr()

        def f8_rows(w, q_rows=0):
            """(pointer to e4m3 [N, K], pointer to fp32 row scales) of a linear weight for the mixed bf16 / fp8 path; the q rows carry
            the softmax exponent factor like to_qkv_qs when that is in use."""
            if not use_f8:
                return None, None
            t = w.detach().to(device=device, dtype=torch.float32).clone()
            if q_rows and q_scale is not None:
                t[:q_rows] *= q_scale
            t = t.contiguous()
            q = torch.empty(t.shape, dtype=torch.uint8, device=device)
            sc = torch.empty(t.shape[0], dtype=torch.float32, device=device)
            _lib.check(_lib.lib().ttv_quant_rows_fp8(t.data_ptr(), _lib.TTV_F32, t.shape[1], None, 0.0, q.data_ptr(), t.shape[1], sc.data_ptr(),
                                                     t.shape[0], t.shape[1], _lib.stream_ptr(device)), "ttv_quant_rows_fp8")
            keep.extend([t, q, sc])
            return q.data_ptr(), sc.data_ptr()

        def folded_tensor(w, g, q_rows=0):
            t = (w.detach().to(device=device, dtype=torch.float32) * g.detach().to(device=device, dtype=torch.float32)[None, :])
            if q_rows and q_scale is not None:
                t[:q_rows] *= q_scale
            return t.to(dtype).contiguous()

        def folded(w, g, q_rows=0):
            """w * gain[None, :] in fp32, then the compute dtype: lets the GEMM absorb the pre-norm."""
            # split-bf16 towers, opt-in (TTV_SPLIT3_FOLD=1): the folded matrix as a split image, the GEMM scales its rows by the pre-norm's
            # rstd - 8 launches less per tiny encoder (+3 % throughput), but the measured max |pre-rounding FSQ value error| on the
            # benchmark fixture goes from 5.4e-4 to 7.3e-4 of the 1e-3 the index guarantee rests on, so the default keeps the norms apart
            if self.f32_split3 and os.environ.get("TTV_SPLIT3_FOLD", "0") == "1":
                t = (w.detach().to(device=device, dtype=torch.float32) * g.detach().to(device=device, dtype=torch.float32)[None, :]).contiguous()
                img = torch.empty_like(t)
                _lib.check(_lib.lib().ttv_split3_pack(t.data_ptr(), t.shape[1], img.data_ptr(), t.shape[1], t.shape[0], t.shape[1], _lib.stream_ptr(device)),
                           "ttv_split3_pack")
                keep.extend([t, img])
                return img.data_ptr()
            if not (fold or fold_gen):
                return None
            t = folded_tensor(w, g, q_rows)
            keep.append(t)
            return t.data_ptr()

        def mlp_packed(w12, g, w3, wo, next_attn):
            """Panel images of the fused layer-tail kernel (csrc/ttv_mlp.hip), built on the device by ttv_mlp_pack from
            w12 * gain[None, :], w3, out_proj and - except for the last layer - the NEXT layer's to_qkv * pre_ln gain.
            Returns (pointer, rows of the packed next-layer to_qkv)."""
            if not fold or w3.shape[1] % 32:
                return None, 0
            w12f = folded_tensor(w12, g)
            w3c = w3.detach().to(device=device, dtype=dtype).contiguous()
            woc = wo.detach().to(device=device, dtype=dtype).contiguous()
            wq, rows = None, 0
            if next_attn is not None and next_attn.to_qkv.weight.shape[0] % 64 == 0:
                wq = folded_tensor(next_attn.to_qkv.weight, next_attn.pre_ln.weight, q_rows=tower.width)
                rows = int(wq.shape[0])
            inner = int(w3.shape[1])
            out = torch.empty(_lib.lib().ttv_mlp_pack_bytes(inner, rows), dtype=torch.uint8, device=device)
            _lib.check(_lib.lib().ttv_mlp_pack(w12f.data_ptr(), w3c.data_ptr(), woc.data_ptr(), wq.data_ptr() if wq is not None else None,
                                               rows, inner, tower.width, _lib.TTV_BF16, out.data_ptr(), _lib.stream_ptr(device)),
                       "mlp_pack")
            keep.extend([w12f, w3c, woc, wq, out])    # stream-ordered: inputs stay alive with the pack
            return out.data_ptr(), rows

        perm = tower._patch_perm_on(device)
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
            pack_ptr, pack_rows = mlp_packed(f.w12.weight, f.norm.weight, f.w3.weight, a.out_proj.weight,
                                             ml.attn_layer[i + 1] if i + 1 < n else None)
            if use_mx:
                qkv8, qkv8s, qkvmx = f8_mx(a.to_qkv.weight, a.pre_ln.weight, q_rows=tower.width)
                w128, w128s, w12mx = f8_mx(f.w12.weight, f.norm.weight)
                wo8, wo8s, womx = f8_mx(a.out_proj.weight)
                w38, w38s, w3mx = f8_mx(f.w3.weight)
            else:
                qkv8, qkv8s = f8_rows(a.to_qkv.weight, q_rows=tower.width)
                w128, w128s = f8_rows(f.w12.weight)
                qkvmx = w12mx = wo8 = wo8s = womx = w38 = w38s = w3mx = None
            self.layers[i] = _lib.LayerWeights(
                to_qkv_f8=qkv8, to_qkv_f8_scale=qkv8s, w12_f8=w128, w12_f8_scale=w128s,
                to_qkv_mx=qkvmx, w12_mx=w12mx, out_proj_f8=wo8, out_proj_f8_scale=wo8s, out_proj_mx=womx,
                w3_f8=w38, w3_f8_scale=w38s, w3_mx=w3mx,
                pre_ln=gain(a.pre_ln.weight), to_qkv=lin(a.to_qkv.weight, True), out_proj=lin(a.out_proj.weight, True),
                ffd_norm=gain(f.norm.weight), w12=lin(f.w12.weight, True), w3=lin(f.w3.weight, True),
                attn_post_ln=gain(ml.attn_post_ln[i - 1].weight) if i > 0 else None,
                ffd_post_ln=gain(ml.ffd_post_ln[i - 1].weight) if i > 0 else None,
                to_qkv_pn=folded(a.to_qkv.weight, a.pre_ln.weight, q_rows=tower.width), w12_pn=folded(f.w12.weight, f.norm.weight),
                mlp_pack=pack_ptr, mlp_pack_qkv_rows=pack_rows, qkv_q_prescaled=self.q_prescaled, to_qkv_qs=lin_qs(a.to_qkv.weight))
        self.struct = _lib.TowerWeights(
            proj_in_w=lin(w_in, tower.kind == _lib.TTV_ENCODER), proj_in_b=lin(tower.proj_in.bias), mask_token=gain(tower.mask_token),
            ln_pre_t=gain(tower.ln_pre_t.weight), ln_pre_p=gain(tower.ln_pre_p.weight), ln_post=gain(tower.ln_post.weight),
            proj_out_w=lin(w_out, tower.kind == _lib.TTV_DECODER), proj_out_b=lin(b_out), layers=self.layers,
            f32_split3=1 if self.f32_split3 else 0,
            proj_out_pn=folded(w_out, tower.ln_post.weight) if (tower.kind == _lib.TTV_DECODER and fold) else None)
        self.keep = keep
        self.kind, self.n_layers = tower.kind, n
        self._t = None
        if self.device.type == "cuda":
            self.build_stream = torch.cuda.current_stream(self.device)
            self.ready = torch.cuda.Event()
            self.ready.record(self.build_stream)

    def use_on_current_stream(self) -> None:
        """Called by every forward: a stream other than the one that built the pack waits for the build (casts, folds, the
        ttv_mlp_pack kernel) before its kernels read the buffers, and is remembered as a reader (see _Tower._retire_pack)."""
        if self.ready is None:
            return
        cur = torch.cuda.current_stream(self.device)
        if cur != self.build_stream and cur.cuda_stream not in self.reader_streams:
            cur.wait_event(self.ready)
        self.reader_streams[cur.cuda_stream] = cur

    def transposed(self) -> "_lib.TowerWeightsT":
        """W^T copies for the data-gradient GEMMs of the backward pass (built on first use, per weight version)."""
        if self._t is None:
            lt = self.lin_tensors   # per layer: to_qkv, out_proj, w12, w3 ; then proj_in_w, proj_in_b, proj_out_w, proj_out_b
            arr = (_lib.LayerWeightsT * self.n_layers)()
            keep = []

            def tr(t):
                u = t.t().contiguous()
                keep.append(u)
                return u.data_ptr()
            for i in range(self.n_layers):
                q, o, w12, w3 = lt[4 * i: 4 * i + 4]
                arr[i] = _lib.LayerWeightsT(to_qkv_t=tr(q), out_proj_t=tr(o), w12_t=tr(w12), w3_t=tr(w3))
            proj_in_w, proj_out_w = lt[4 * self.n_layers], lt[4 * self.n_layers + 2]
            st = _lib.TowerWeightsT(proj_in_t=tr(proj_in_w) if self.kind == _lib.TTV_ENCODER else None,
                                    proj_out_t=tr(proj_out_w) if self.kind == _lib.TTV_DECODER else None, layers=arr)
            self._t = (st, arr, keep)
        return self._t[0]


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

    def forward_z(self, videos, token_counts, grids=None) -> torch.Tensor:
        """fp32 pre-quantisation tokens [sum(K), out_channels]; differentiable when grad is enabled (training step)."""
        if self._wants_grad(*videos):
            counts = host_ints(token_counts)
            pix = [tuple(v.shape[1:]) for v in videos] if grids is None else [tuple(int(x) for x in g) for g in host_ints(grids)]
            return _EncoderTrainFn.apply(self, counts, pix, len(videos), *videos, *self._param_list())
        return self.run(videos, token_counts, grids, None, want_z=True)["z"]

    def forward(self, videos, token_counts, grids=None):
        return self.forward_z(videos, token_counts, grids).to(videos[0].dtype)


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
        if self._wants_grad(tokens):
            counts = host_ints(token_counts)
            pix = [tuple(int(v) for v in g) for g in host_ints(grids)]
            outs = _DecoderTrainFn.apply(self, counts, pix, tokens, *self._param_list())
            return list(outs)
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


# ------------------------------------------------------------------------------------------------------------------
# autograd: tape-recording forward + hand-written HIP backward (csrc/ttv_train.hip, ttv_bwd.hip)
# ------------------------------------------------------------------------------------------------------------------
def _train_ctx(tower: _Tower, pix, counts, device, dtype):
    if getattr(tower, "f32_split3", False):
        raise RuntimeError("f32_split3 (split-bf16 arithmetic) is an inference mode: the training towers run plain bf16 / fp32")
    plan = tower._plan(pix, counts, device)
    dims = tower._dims(_lib.dtype_code(dtype))
    batch = plan.batch_for(tower.heads[0], tower.heads[1])
    pack = tower._packed(dtype, device)
    lib = _lib.lib()
    tape = torch.empty(int(lib.ttv_tower_tape_bytes(C.byref(dims), C.byref(batch))), dtype=torch.uint8, device=device)
    return plan, dims, batch, pack, tape


class _EncoderTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tower, counts, pix, n_clips, *tensors):
        clips = [c.contiguous() for c in tensors[:n_clips]]
        device, dtype = clips[0].device, clips[0].dtype
        plan, dims, batch, pack, tape = _train_ctx(tower, pix, counts, device, dtype)
        z = torch.empty((plan.sum_tokens, tower.token_size), dtype=torch.float32, device=device)
        rc = _lib.lib().ttv_encoder_forward_train(C.byref(dims), C.byref(pack.struct), C.byref(batch), _lib.ptr_array(clips),
                                                  z.data_ptr(), tape.data_ptr(), tape.numel(), _lib.stream_ptr(device))
        _lib.check(rc, "ttv_encoder_forward_train")
        ctx.tower, ctx.plan, ctx.dims, ctx.batch, ctx.pack, ctx.tape = tower, plan, dims, batch, pack, tape
        ctx.clips = clips
        ctx.n_clips = n_clips
        ctx.clip_grad = [bool(t.requires_grad) for t in tensors[:n_clips]]
        ctx.params = list(tensors[n_clips:])
        return z

    @staticmethod
    def backward(ctx, dz):
        tower, device = ctx.tower, dz.device
        lib = _lib.lib()
        ws = torch.empty(int(lib.ttv_tower_bwd_workspace_bytes(C.byref(ctx.dims), C.byref(ctx.batch))), dtype=torch.uint8, device=device)
        dclips = None
        if any(ctx.clip_grad):
            dclips = [torch.empty_like(c) for c in ctx.clips]
        # every parameter frozen (the generator step through the discriminator, loss_module.py:144-151): inputs-only backward
        frozen = dclips is not None and not any(p.requires_grad for p in ctx.params)
        if not frozen:
            flat, gstruct, _lay = tower._grad_buffers(device)
        rc = lib.ttv_encoder_backward(C.byref(ctx.dims), C.byref(ctx.pack.struct), C.byref(ctx.pack.transposed()), C.byref(ctx.batch),
                                      dz.contiguous().float().data_ptr(), ctx.tape.data_ptr(), None if frozen else C.byref(gstruct),
                                      _lib.ptr_array(dclips) if dclips else None, ws.data_ptr(), ws.numel(), _lib.stream_ptr(device))
        _lib.check(rc, "ttv_encoder_backward")
        by_id = {} if frozen else tower._reduce_and_finish(flat)
        clip_grads = [dclips[i] if (dclips and ctx.clip_grad[i]) else None for i in range(ctx.n_clips)]
        param_grads = [by_id.get(id(p)) if p.requires_grad else None for p in ctx.params]
        ctx.tape = None
        return (None, None, None, None, *clip_grads, *param_grads)


class _DecoderTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tower, counts, pix, tokens, *params):
        device, dtype = tokens.device, tokens.dtype
        tokens = tokens.contiguous()
        plan, dims, batch, pack, tape = _train_ctx(tower, pix, counts, device, dtype)
        sizes = [tower.out_channels * t * h * w for (t, h, w) in pix]
        outs = [torch.empty((tower.out_channels, t, h, w), dtype=dtype, device=device) for (t, h, w) in pix]
        pd = tower.out_channels * math.prod(tower.patch)
        ws = torch.empty(plan.sum_patches * pd, dtype=dtype, device=device)
        rc = _lib.lib().ttv_decoder_forward_train(C.byref(dims), C.byref(pack.struct), C.byref(batch), tokens.data_ptr(),
                                                  _lib.ptr_array(outs), tape.data_ptr(), tape.numel(), ws.data_ptr(),
                                                  ws.numel() * ws.element_size(), _lib.stream_ptr(device))
        _lib.check(rc, "ttv_decoder_forward_train")
        ctx.tower, ctx.plan, ctx.dims, ctx.batch, ctx.pack, ctx.tape = tower, plan, dims, batch, pack, tape
        ctx.tokens = tokens
        ctx.tokens_grad = bool(tokens.requires_grad)
        ctx.params = list(params)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dclips):
        tower, device = ctx.tower, ctx.tokens.device
        lib = _lib.lib()
        flat, gstruct, _lay = tower._grad_buffers(device)
        ws = torch.empty(int(lib.ttv_tower_bwd_workspace_bytes(C.byref(ctx.dims), C.byref(ctx.batch))), dtype=torch.uint8, device=device)
        shapes = [(tower.out_channels, t, h, w) for (t, h, w) in ctx.plan.pixel_grids]
        dcl = [(g if g is not None else torch.zeros(sh, dtype=ctx.tokens.dtype, device=device)).to(ctx.tokens.dtype).contiguous()
               for g, sh in zip(dclips, shapes)]
        dcodes = torch.empty((ctx.plan.sum_tokens, tower.token_size), dtype=torch.float32, device=device)
        rc = lib.ttv_decoder_backward(C.byref(ctx.dims), C.byref(ctx.pack.struct), C.byref(ctx.pack.transposed()), C.byref(ctx.batch),
                                      ctx.tokens.data_ptr(), _lib.ptr_array(dcl), ctx.tape.data_ptr(), C.byref(gstruct),
                                      dcodes.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(device))
        _lib.check(rc, "ttv_decoder_backward")
        by_id = tower._reduce_and_finish(flat)
        param_grads = [by_id.get(id(p)) if p.requires_grad else None for p in ctx.params]
        ctx.tape = None
        return (None, None, None, dcodes.to(ctx.tokens.dtype) if ctx.tokens_grad else None, *param_grads)
