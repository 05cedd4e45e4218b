"""Per-batch metadata, computed on the host from Python ints and uploaded once per distinct batch shape.

Replaces the reference's device-side bookkeeping and the host syncs it causes (model/base/blocks.py:80-88,
154-162: grids//patch, cu_seqlens, bool mask by repeat_interleave; model/base/rope.py:57-71: per-sample Python
loop).  Row layout of a packed batch: clip b owns rows [cu[b], cu[b+1]); its K_b latent tokens come first, then
its P_b patch tokens in (t,h,w) raster order (blocks.py:85-86).

The RoPE table follows rope.py:40-54 exactly: fp64 `theta**linspace(0,1,F) * pi/2` times the fp32 position ids,
`torch.polar` in fp64, then cast to fp32 (the reference casts to complex64 at apply time, rope.py:24).
"""
from __future__ import annotations

import ctypes as C
import functools
import math
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib

QBLOCK = 128  # query rows per attention workgroup (csrc/ttv_attn.hip QB)


@functools.lru_cache(maxsize=256)
def _rope_clip_table(grid: Tuple[int, ...], k: int, head_dim: int, theta: float = 10000.0) -> np.ndarray:
    """fp32 [K+P, 64] = cos[32] | sin[32] for one clip; pairs beyond 3*F keep (1, 0) (rope.py:24 leaves them alone)."""
    nd = len(grid)
    f = head_dim // (2 * nd)
    inv = torch.pow(theta, torch.linspace(0.0, 1.0, f, dtype=torch.float64)) * torch.pi / 2.0
    tok = torch.arange(k, dtype=torch.float32).unsqueeze(-1).expand(-1, nd)
    coords = [torch.arange(int(g), dtype=torch.float32) for g in grid]
    gid = torch.cartesian_prod(*coords) + k
    if gid.dim() == 1:
        gid = gid.unsqueeze(-1)
    ids = torch.cat([tok, gid], dim=0)
    ang = (inv.view(1, -1, 1) * ids.to(torch.float64).unsqueeze(-2)).reshape(ids.shape[0], -1)
    fc = torch.polar(torch.ones(1, dtype=torch.float64), ang)
    half = head_dim // 2
    out = np.zeros((ids.shape[0], 2 * half), dtype=np.float32)
    out[:, :half] = 1.0
    out[:, : nd * f] = fc.real.to(torch.float32).numpy()
    out[:, half: half + nd * f] = fc.imag.to(torch.float32).numpy()
    return out


class BatchPlan:
    """Device tables + the ttv_batch struct for one (clip shapes, token counts) combination."""

    def __init__(self, pixel_grids: Sequence[Sequence[int]], token_counts: Sequence[int], patch: Sequence[int],
                 device: torch.device, head_dim: int = 64):
        if head_dim != 64:
            raise ValueError("head_dim is fixed at 64 (reference model/base/utils.py:8)")
        self.device = torch.device(device)
        self.patch = tuple(int(p) for p in patch)
        self.pixel_grids = [tuple(int(v) for v in g) for g in pixel_grids]
        self.token_counts = [int(k) for k in token_counts]
        if len(self.pixel_grids) != len(self.token_counts) or not self.pixel_grids:
            raise ValueError("need one token count per clip and at least one clip")
        grids, sizes = [], []
        for pg in self.pixel_grids:
            if len(pg) != len(self.patch) or any(v <= 0 or v % p for v, p in zip(pg, self.patch)):
                raise ValueError(f"clip shape {pg} is not a positive multiple of patch size {self.patch}")
            g = tuple(v // p for v, p in zip(pg, self.patch))
            grids.append(g)
            sizes.append(math.prod(g))
        if any(k < 0 for k in self.token_counts):
            raise ValueError("token counts must be >= 0")
        self.grids, self.grid_sizes = grids, sizes
        B = len(grids)
        cu = [0]
        for k, p in zip(self.token_counts, sizes):
            if k + p <= 0:
                raise ValueError("empty sequence")
            cu.append(cu[-1] + k + p)
        self.cu_seqlens = cu
        self.total_rows = cu[-1]
        self.sum_tokens = sum(self.token_counts)
        self.sum_patches = sum(sizes)
        self.max_seqlen = max(k + p for k, p in zip(self.token_counts, sizes))

        latent_rows = np.concatenate([np.arange(cu[b], cu[b] + self.token_counts[b], dtype=np.int32) for b in range(B)])
        patch_rows = np.concatenate([np.arange(cu[b] + self.token_counts[b], cu[b + 1], dtype=np.int32) for b in range(B)])
        desc = np.zeros((B, 8), dtype=np.int32)
        pbase = 0
        for b in range(B):
            T, H, W = self.pixel_grids[b]
            desc[b] = (T, H, W, grids[b][0], grids[b][1], grids[b][2], pbase, 3)
            pbase += sizes[b]
        blocks64 = np.asarray([(b, r0) for b in range(B) for r0 in range(0, cu[b + 1] - cu[b], 64)], dtype=np.int32).reshape(-1, 2)
        row_seq = np.concatenate([np.full(cu[b + 1] - cu[b], b, dtype=np.int32) for b in range(B)])
        self.n_blocks64 = int(blocks64.shape[0])
        parts = [np.asarray(cu, dtype=np.int32), latent_rows, patch_rows, desc.reshape(-1), blocks64.reshape(-1), row_seq]
        offs, total = [], 0
        for p in parts:
            offs.append(total)
            total += (p.size + 3) // 4 * 4      # keep every table 16-byte aligned
        host = np.zeros(total, dtype=np.int32)
        for o, p in zip(offs, parts):
            host[o:o + p.size] = p
        self.int_tables = torch.from_numpy(host).to(self.device, non_blocking=False)
        rope = np.concatenate([_rope_clip_table(g, k, head_dim) for g, k in zip(grids, self.token_counts)], axis=0)
        self.rope_cs = torch.from_numpy(rope).to(self.device)

        base = self.int_tables.data_ptr()
        self._base_fields = dict(
            n_clips=B, total_rows=self.total_rows, sum_tokens=self.sum_tokens, sum_patches=self.sum_patches,
            max_patches_per_clip=max(sizes),
            cu_seqlens=base + 4 * offs[0], latent_rows=base + 4 * offs[1], patch_rows=base + 4 * offs[2],
            clip_desc=base + 4 * offs[3], rope_cs=self.rope_cs.data_ptr(),
            blocks64=base + 4 * offs[4], row_seq=base + 4 * offs[5], n_blocks64=self.n_blocks64)
        self._offs = offs
        self._attn = {}

    def attention_table(self, q_heads: int, kv_heads: int) -> torch.Tensor:
        """int32 [n,4] attention work table (sequence, first query row, q-head, 0) for this batch, XCD-aware.

        Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md: blocks b and b+8 share an XCD - a speed
        assumption only, never correctness).  All blocks of one (sequence, kv-head) unit re-read the same K/V, so units
        are distributed over 8 lists (greedy by block count) and the lists are interleaved: entry i goes to list i % 8.
        Shorter lists are padded with sequence = -1 entries (the kernel returns immediately)."""
        key = (int(q_heads), int(kv_heads))
        t = self._attn.get(key)
        if t is None:
            rep = q_heads // kv_heads
            units = []
            for b in range(len(self.grids)):
                s = self.cu_seqlens[b + 1] - self.cu_seqlens[b]
                for kvh in range(kv_heads):
                    units.append([(b, q0, kvh * rep + r, 0) for q0 in range(0, s, QBLOCK) for r in range(rep)])
            lists = [[] for _ in range(8)]
            for u in sorted(units, key=len, reverse=True):
                min(lists, key=len).extend(u)
            depth = max(len(l) for l in lists)
            table = np.full((depth, 8, 4), -1, dtype=np.int32)
            for x, l in enumerate(lists):
                if l:
                    table[: len(l), x, :] = np.asarray(l, dtype=np.int32)
            # drop trailing all-padding rows only (interior padding keeps the i % 8 alignment)
            flat = table.reshape(-1, 4)
            last = int(np.max(np.nonzero(flat[:, 0] >= 0)[0])) + 1
            t = torch.from_numpy(np.ascontiguousarray(flat[:last])).to(self.device)
            self._attn[key] = t
        return t

    def batch_for(self, q_heads: int, kv_heads: int) -> "_lib.Batch":
        """ttv_batch struct whose attention work table matches the tower's head counts."""
        t = self.attention_table(q_heads, kv_heads)
        return _lib.Batch(n_qblocks=int(t.shape[0]), qblocks=t.data_ptr(), **self._base_fields)

    # views used by tests that call single ops
    def table(self, i: int, n: int) -> torch.Tensor:
        return self.int_tables[self._offs[i]: self._offs[i] + n]

    @property
    def cu_dev(self):
        return self.table(0, len(self.cu_seqlens))

    @property
    def latent_rows_dev(self):
        return self.table(1, self.sum_tokens)

    @property
    def patch_rows_dev(self):
        return self.table(2, self.sum_patches)

    @property
    def clip_desc_dev(self):
        return self.table(3, 8 * len(self.grids))



_plan_cache = {}
_PLAN_CACHE_MAX = 64


def get_plan(pixel_grids, token_counts, patch, device) -> BatchPlan:
    key = (tuple(tuple(int(v) for v in g) for g in pixel_grids), tuple(int(k) for k in token_counts),
           tuple(int(p) for p in patch), str(torch.device(device)))
    plan = _plan_cache.get(key)
    if plan is None:
        if len(_plan_cache) >= _PLAN_CACHE_MAX:
            _plan_cache.pop(next(iter(_plan_cache)))
        plan = BatchPlan(key[0], key[1], key[2], device)
        _plan_cache[key] = plan
    return plan


def host_ints(v) -> List[int]:
    """token_counts / grids as Python ints.  A device tensor costs one host sync (the reference syncs several times
    per tower for the same information); pass lists or CPU tensors to avoid it."""
    if isinstance(v, torch.Tensor):
        return v.detach().to("cpu").tolist()
    return [x.tolist() if hasattr(x, "tolist") else x for x in v]
