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
    def __init__(self, config):
        super().__init__()
        self.config = config
        conf = config.tokenizer.model
        levels = list(conf.fsq_levels)
        token_size = len(levels)
        patch = tuple(int(p) for p in conf.patch_size)
        self.encoder = TiTokEncoder(model_size=conf.encoder_size, patch_size=patch, in_channels=3, out_channels=token_size)
        self.quantize = FSQ(levels=levels)
        self.decoder = TiTokDecoder(model_size=conf.decoder_size, patch_size=patch, in_channels=token_size, out_channels=3)
        self.apply(init_weights)
        self.last_bounded = None   # fp32 FSQ pre-rounding values of the last encode(want_bounded=True)

    def encode(self, x, token_counts, grids=None, split_indices=False, want_bounded=False):
        counts = host_ints(token_counts)
        if self.encoder._wants_grad(*x):
            # training step (train.py:65-83): differentiable towers (tape + HIP backward) and straight-through FSQ
            z = self.encoder.forward_z(x, counts, grids)                 # fp32, carries the autograd graph
            codes, x_dict = self.quantize(z)
            self.last_bounded = None
            if split_indices:
                x_dict["indices"] = torch.split(x_dict["indices"], counts, dim=0)
            return codes.to(x[0].dtype), x_dict
        out = self.encoder.run(x, counts, grids, self.quantize.params, want_z=False, want_bounded=want_bounded)
        self.last_bounded = out["bounded"]
        indices = out["indices"]
        if split_indices:
            indices = torch.split(indices, counts, dim=0)
        return out["codes"], {"indices": indices}

    def decode_indices(self, indices, grids, token_counts=None):
        if token_counts is None:
            assert type(indices) in [list, tuple]
            token_counts = [int(t.shape[0]) for t in indices]
            indices = torch.cat(list(indices), dim=0)
        # reference: decoder parameter dtype (titok.py:61); under autocast (Lightning bf16-mixed, fp32 masters)
        # the towers compute in the autocast dtype, so follow it
        dtype = next(self.decoder.parameters()).dtype
        if torch.is_autocast_enabled():
            dtype = torch.get_autocast_gpu_dtype()
        x_q = self.quantize.indices_to_codes(indices, dtype=dtype)
        return self.decoder(x_q, token_counts, grids)

    def decode(self, x, token_counts, grids):
        return self.decoder(x, token_counts, grids)

    def forward(self, x, token_counts):
        grids = [tuple(im.shape[1:]) for im in x]     # host ints; the reference builds a device tensor (titok.py:70)
        counts = host_ints(token_counts)
        x_q, out_dict = self.encode(x, counts, grids)
        recon = self.decode(x_q, counts, grids)
        return recon, out_dict
