"""ctypes binding of libtitok_hip.so (C-ABI declared in include/titok_hip.h).

The library is loaded on first use.  If it is missing the call raises: there is no eager-PyTorch or CPU
fallback behind these entry points.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

TTV_BF16, TTV_F32 = 0, 1
TTV_ENCODER, TTV_DECODER = 0, 1
TTV_MAX_FSQ = 8
TTV_MAX_TOKEN = 64
TTV_MAX_CLIPS_PER_LAUNCH = 64

_HERE = os.path.dirname(os.path.abspath(__file__))
# TTV_LIB_PATH: diagnostics only (an instrumented build of the same sources, tools/attn_stamps.py)
LIB_PATH = os.environ.get("TTV_LIB_PATH") or os.path.join(_HERE, "libtitok_hip.so")

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class FsqParams(C.Structure):
    _fields_ = [("n", i32), ("levels", i32 * TTV_MAX_FSQ), ("basis", i32 * TTV_MAX_FSQ),
                ("half_l", f32 * TTV_MAX_FSQ), ("offset", f32 * TTV_MAX_FSQ), ("shift", f32 * TTV_MAX_FSQ),
                ("half_width", f32 * TTV_MAX_FSQ)]


class TowerDims(C.Structure):
    _fields_ = [("kind", i32), ("dtype", i32), ("width", i32), ("layers", i32), ("q_heads", i32), ("kv_heads", i32),
                ("head_dim", i32), ("inner", i32), ("patch_t", i32), ("patch_h", i32), ("patch_w", i32),
                ("pix_channels", i32), ("token_size", i32), ("eps", f32), ("alpha", f32)]


class NextQkv(C.Structure):
    _fields_ = [("qkv", vp), ("ld", i32), ("rope_cs", vp), ("rows", i32), ("rope_q_end", i32), ("rope_k_begin", i32), ("rope_k_end", i32)]


class LayerWeights(C.Structure):
    _fields_ = [("pre_ln", vp), ("to_qkv", vp), ("out_proj", vp), ("ffd_norm", vp), ("w12", vp), ("w3", vp),
                ("attn_post_ln", vp), ("ffd_post_ln", vp), ("to_qkv_pn", vp), ("w12_pn", vp), ("mlp_pack", vp), ("mlp_pack_qkv_rows", i32), ("qkv_q_prescaled", i32),
                ("to_qkv_qs", vp), ("to_qkv_f8", vp), ("to_qkv_f8_scale", vp), ("w12_f8", vp), ("w12_f8_scale", vp),
                ("to_qkv_mx", vp), ("w12_mx", vp), ("out_proj_f8", vp), ("out_proj_f8_scale", vp), ("out_proj_mx", vp),
                ("w3_f8", vp), ("w3_f8_scale", vp), ("w3_mx", vp)]


class TowerWeights(C.Structure):
    _fields_ = [("proj_in_w", vp), ("proj_in_b", vp), ("mask_token", vp), ("ln_pre_t", vp), ("ln_pre_p", vp),
                ("ln_post", vp), ("proj_out_w", vp), ("proj_out_b", vp), ("layers", C.POINTER(LayerWeights)), ("proj_out_pn", vp),
                ("f32_split3", i32)]


class Batch(C.Structure):
    _fields_ = [("n_clips", i32), ("total_rows", i32), ("sum_tokens", i32), ("sum_patches", i32),
                ("max_patches_per_clip", i32), ("n_qblocks", i32), ("cu_seqlens", vp), ("latent_rows", vp),
                ("patch_rows", vp), ("clip_desc", vp), ("qblocks", vp), ("rope_cs", vp), ("blocks64", vp), ("row_seq", vp),
                ("n_blocks64", i32), ("qblocks_paired", i32), ("qblocks_all_full", i32), ("items64", vp), ("n_items64", i32),
                ("rope_ids", vp), ("rope_base", vp), ("qblocks_latent", vp), ("n_qblocks_latent", i32), ("qblocks_patch", vp),
                ("n_qblocks_patch", i32)]


class LayerWeightsT(C.Structure):
    _fields_ = [("to_qkv_t", vp), ("out_proj_t", vp), ("w12_t", vp), ("w3_t", vp)]


class TowerWeightsT(C.Structure):
    _fields_ = [("proj_in_t", vp), ("proj_out_t", vp), ("layers", C.POINTER(LayerWeightsT))]


class LayerGrads(C.Structure):
    _fields_ = [("pre_ln", vp), ("to_qkv", vp), ("out_proj", vp), ("ffd_norm", vp), ("w12", vp), ("w3", vp),
                ("attn_post_ln", vp), ("ffd_post_ln", vp)]


class TowerGrads(C.Structure):
    _fields_ = [("proj_in_w", vp), ("proj_in_b", vp), ("mask_token", vp), ("ln_pre_t", vp), ("ln_pre_p", vp), ("ln_post", vp),
                ("proj_out_w", vp), ("proj_out_b", vp), ("layers", C.POINTER(LayerGrads)), ("layer_done_events", C.POINTER(vp))]


_lib = None
_lock = threading.Lock()

# every symbol include/titok_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "ttv_error_string": (C.c_char_p, []),
    "ttv_version": (C.c_int, []),
    "ttv_fsq_forward": (C.c_int, [C.POINTER(FsqParams), vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp]),
    "ttv_fsq_indices_to_codes": (C.c_int, [C.POINTER(FsqParams), vp, C.c_int, vp, C.c_int, vp]),
    "ttv_vq_codebook_norms": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "ttv_vq_workspace_bytes": (C.c_int64, [C.c_int]),
    "ttv_vq_l2_argmin": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int64, vp]),
    "ttv_vq_lookup": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp]),
    "ttv_vq_lookup_backward": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp]),
    "ttv_quant_rows_fp8": (C.c_int, [vp, C.c_int, C.c_int, vp, f32, vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "ttv_linear_fp8": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]),
    "ttv_split3_pack": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "ttv_linear_split3": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ttv_mx_scale_bytes_per_row": (C.c_int64, [C.c_int]),
    "ttv_quant_mx_fp8": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp]),
    "ttv_linear_fp8_mx": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int,
                                    vp, C.c_int, f32, vp]),
    "ttv_rmsnorm": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, f32, vp]),
    "ttv_rope_apply": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "ttv_linear": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ttv_linear_qkv_rope": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "ttv_linear_geglu": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ttv_linear_residual": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, f32, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, vp]),
    "ttv_linear_residual_norm": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, f32, vp, f32, vp, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, vp]),
    "ttv_mlp_pack_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "ttv_mlp_pack": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "ttv_layer_tail_fused": (C.c_int, [vp, C.c_int, vp, f32, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, f32, f32, C.c_int, C.c_int,
                                       C.c_int, vp, vp]),
    "ttv_mlp_fused": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, f32, f32, C.c_int, C.c_int, C.c_int, vp]),
    "ttv_fill_const_rows": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, f32, vp]),
    "ttv_decoder_embed": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, f32, vp]),
    "ttv_attention": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ttv_attention64": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ttv_patch_gather": (C.c_int, [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int,
                                   C.c_int, C.c_int, vp]),
    "ttv_patch_scatter": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp),
                                    C.c_int, C.c_int, vp]),
    "ttv_tower_workspace_bytes": (i64, [C.POINTER(TowerDims), C.POINTER(Batch)]),
    "ttv_encoder_forward": (C.c_int, [C.POINTER(TowerDims), C.POINTER(TowerWeights), C.POINTER(Batch), C.POINTER(vp),
                                      C.POINTER(FsqParams), vp, vp, vp, vp, vp, i64, vp]),
    "ttv_decoder_forward": (C.c_int, [C.POINTER(TowerDims), C.POINTER(TowerWeights), C.POINTER(Batch), vp, C.POINTER(vp),
                                      vp, i64, vp]),
    "ttv_tower_tape_bytes": (i64, [C.POINTER(TowerDims), C.POINTER(Batch)]),
    "ttv_tower_bwd_workspace_bytes": (i64, [C.POINTER(TowerDims), C.POINTER(Batch)]),
    "ttv_encoder_forward_train": (C.c_int, [C.POINTER(TowerDims), C.POINTER(TowerWeights), C.POINTER(Batch), C.POINTER(vp), vp, vp,
                                            i64, vp]),
    "ttv_encoder_backward": (C.c_int, [C.POINTER(TowerDims), C.POINTER(TowerWeights), C.POINTER(TowerWeightsT), C.POINTER(Batch), vp,
                                       vp, C.POINTER(TowerGrads), C.POINTER(vp), vp, i64, vp]),
    "ttv_decoder_forward_train": (C.c_int, [C.POINTER(TowerDims), C.POINTER(TowerWeights), C.POINTER(Batch), vp, C.POINTER(vp), vp,
                                            i64, vp, i64, vp]),
    "ttv_decoder_backward": (C.c_int, [C.POINTER(TowerDims), C.POINTER(TowerWeights), C.POINTER(TowerWeightsT), C.POINTER(Batch), vp,
                                       C.POINTER(vp), vp, C.POINTER(TowerGrads), vp, vp, i64, vp]),
    "ttv_fsq_backward": (C.c_int, [C.POINTER(FsqParams), vp, vp, C.c_int, vp, C.c_int, vp]),
    "ttv_rmsnorm_backward_chain": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, C.c_float, vp, C.c_int,
                                              C.c_int, C.c_int, C.c_float, C.c_int, vp]),
    "ttv_opt_grad_sumsq": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "ttv_opt_adamw_step": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int] + [C.c_float] * 8 + [vp, vp]),
    "ttv_linear_wgrad_workspace_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "ttv_linear_wgrad": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int64, vp]),
    "ttv_rmsnorm_backward": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, C.c_int, C.c_int, f32, C.c_int, vp]),
    "ttv_attention_backward": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, C.c_int,
                                         C.c_int, C.c_int, C.c_int, vp, vp]),
    "ttv_attention_lse": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "ttv_codebook_histogram": (C.c_int, [vp, C.c_int, vp, C.c_int, vp]),
    "ttv_rope_table_build": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp]),
    "ttv_l1_loss": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "ttv_clip_from_u8": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "ttv_sq_err_accumulate": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "ttv_debug_set": (C.c_int, [C.c_int]),
    "ttv_debug_stamps": (C.c_int, [vp]),
    "ttv_prof_begin": (C.c_int, [C.c_int, C.c_int]),
    "ttv_prof_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int)]),
}

KERNEL_CLASSES = {"attention": 1, "gemm_qkv": 2, "gemm_geglu": 3, "gemm_resid": 4, "gemm_store": 5, "rmsnorm": 6,
                  "patch": 7, "rows": 8}


def lib():
    """The loaded library (raises RuntimeError when it has not been built)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"titok_video_amd: {LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; "
                        "g.build()'` (or titok_video_amd/csrc/build.sh). There is no non-HIP fallback.")
                handle = C.CDLL(LIB_PATH)
                for name, (res, args) in SYMBOLS.items():
                    fn = getattr(handle, name)
                    fn.restype, fn.argtypes = res, args
                _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ttv_error_string().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return TTV_BF16
    if dt == torch.float32:
        return TTV_F32
    raise TypeError(f"titok_video_amd supports bfloat16 and float32 activations, got {dt}")


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; titok_video_amd runs on MI355X (HIP) only, there is no CPU path")


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def ptr_array(tensors):
    arr = (vp * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
