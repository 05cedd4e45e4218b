"""TiTok: the reference tokenizer facade (model/titok.py:23-74) on the MI355X HIP path.

Same constructor (`config.tokenizer.model.{fsq_levels, encoder_size, decoder_size, patch_size}`), methods, return
types and state-dict keys (`encoder.*`, `decoder.*`; FSQ has no persistent keys).  Differences, on purpose:
  * `token_counts` / `grids` may be Python lists or CPU tensors (no host sync) as well as the reference's device tensors;
  * `encode(..., split_indices=True)` works (it raises TypeError upstream) and returns a tuple
    of per-clip index tensors;
  * the encoder tail and FSQ are one kernel, with the pre-quantisation tokens kept in fp32 (the reference rounds
    them to bf16 under autocast before `z.float()`, fsq.py:128).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..plan import host_ints
from .base.blocks import TiTokDecoder, TiTokEncoder
from .base.utils import init_weights
from .quantizer.fsq import FSQ


class TiTok(nn.Module):
    """encoder -> FSQ -> decoder over lists of [3,T,H,W] clips; parameters live in `encoder.*` and `decoder.*`."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        m = config.tokenizer.model
        fsq_levels = [int(v) for v in m.fsq_levels]
        patch = tuple(int(p) for p in m.patch_size)
        n_code = len(fsq_levels)                       # one latent channel per FSQ level
        towers = dict(encoder=TiTokEncoder(model_size=m.encoder_size, patch_size=patch, in_channels=3, out_channels=n_code),
                      quantize=FSQ(levels=fsq_levels),
                      decoder=TiTokDecoder(model_size=m.decoder_size, patch_size=patch, in_channels=n_code, out_channels=3))
        for name, mod in towers.items():               # registration order = the reference's state-dict order
            setattr(self, name, mod)
        self.apply(init_weights)
        self.last_bounded = None   # fp32 FSQ pre-rounding values of the last encode(want_bounded=True)

    # ---- encode ---------------------------------------------------------------------------------------------------------
    def _encode_differentiable(self, clips, counts, grids, split_indices):
        """Training step (train.py:65-83): tape-recording towers with the HIP backward, straight-through FSQ."""
        z = self.encoder.forward_z(clips, counts, grids)               # fp32, carries the autograd graph
        codes, info = self.quantize(z)
        self.last_bounded = None
        if split_indices:
            info["indices"] = torch.split(info["indices"], counts, dim=0)
        return codes.to(clips[0].dtype), info

    def encode(self, x, token_counts, grids=None, split_indices=False, want_bounded=False):
        counts = host_ints(token_counts)
        if self.encoder._wants_grad(*x):
            return self._encode_differentiable(x, counts, grids, split_indices)
        res = self.encoder.run(x, counts, grids, self.quantize.params, want_z=False, want_bounded=want_bounded)
        self.last_bounded = res["bounded"]
        idx = torch.split(res["indices"], counts, dim=0) if split_indices else res["indices"]
        return res["codes"], {"indices": idx}

    # ---- decode ---------------------------------------------------------------------------------------------------------
    def decode(self, x, token_counts, grids):
        return self.decoder(x, token_counts, grids)

    def decode_indices(self, indices, grids, token_counts=None):
        if token_counts is None:                        # per-clip index tensors (titok.py:54-58)
            assert type(indices) in [list, tuple]
            token_counts = [int(part.shape[0]) for part in indices]
            indices = torch.cat(list(indices), dim=0)
        # reference: decoder parameter dtype (titok.py:61); under autocast (Lightning bf16-mixed, fp32 masters)
        # the towers compute in the autocast dtype, so follow it
        compute_dtype = torch.get_autocast_gpu_dtype() if torch.is_autocast_enabled() else next(self.decoder.parameters()).dtype
        return self.decode(self.quantize.indices_to_codes(indices, dtype=compute_dtype), token_counts, grids)

    def forward(self, x, token_counts):
        pixel_grids = [tuple(clip.shape[1:]) for clip in x]     # host ints; the reference builds a device tensor (titok.py:70)
        counts = host_ints(token_counts)
        codes, info = self.encode(x, counts, pixel_grids)
        return self.decode(codes, counts, pixel_grids), info
