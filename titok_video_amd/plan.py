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
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

QBLOCK = 128  # query rows per attention workgroup (csrc/ttv_attn.hip QB)
ATTN_SLOTS = 1024  # table size below which the last third of every sequence's blocks become half items (set when the kernel held 4 blocks per CU; it holds 3 = 768 since round 2 - the rule was re-measured, not re-derived: see attention_table)


@functools.lru_cache(maxsize=8)
def _rope_base_table(head_dim: int, nd: int, n_ids: int, theta: float = 10000.0):
    """fp32 cos/sin of inv_freq[f] * n for every integer position id n < n_ids, evaluated exactly like the reference
    (rope.py:40-54: fp64 `theta**linspace(0,1,F) * pi/2`, fp64 product with the id, torch.polar, cast to fp32).  Position
    ids are small integers (latent index, or grid coordinate + token count), so every table entry the reference computes
    per row is one of these n_ids * F values - rows are gathered from here instead of re-evaluating fp64 trigonometry."""
    f = head_dim // (2 * nd)
    inv = torch.pow(theta, torch.linspace(0.0, 1.0, f, dtype=torch.float64)) * torch.pi / 2.0
    ids = torch.arange(n_ids, dtype=torch.float32).to(torch.float64)
    ang = ids.view(-1, 1) * inv.view(1, -1)                       # [n_ids, F]; same fp64 product as inv * id
    fc = torch.polar(torch.ones(1, dtype=torch.float64), ang)
    return fc.real.to(torch.float32).numpy(), fc.imag.to(torch.float32).numpy()


_copy_streams = {}
_pinned_stage = {}


def _upload(host: np.ndarray, device) -> torch.Tensor:
    """Host table -> device, complete on return, without waiting for compute.  A plain `.to(device)` from pageable memory is
    ordered after everything queued on the current stream, i.e. it waits for the previous batch's forward, which serialises the
    host with the GPU when every batch needs a new plan (ragged batches from a loader).  Here the copy goes through pinned
    memory on a stream of its own and only that copy is waited for (~10 us), so the table can be used from any stream."""
    t = torch.from_numpy(host)
    dev = torch.device(device)
    if dev.type != "cuda":
        return t.to(dev)
    cs = _copy_streams.get(str(dev))
    if cs is None:
        cs = _copy_streams[str(dev)] = torch.cuda.Stream(device=dev)
    # staging through a CACHED pinned buffer: `pin_memory()` per table is a hipHostMalloc (~1 ms; six tables per training step on
    # ragged batches).  The copy is waited for before return, so one buffer per device can be reused by the next call.
    nbytes = t.numel() * t.element_size()
    stage = _pinned_stage.get(str(dev))
    if stage is None or stage.numel() < nbytes:
        stage = _pinned_stage[str(dev)] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
    view = stage[:nbytes].view(t.dtype).view(t.shape)
    # numpy, not Tensor.copy_: above 32 K elements torch's CPU copy is a parallel region of its intra-op pool, and waking that pool
    # (128 threads on a 16-CPU container share) starves the launch path of the whole process (train.limit_host_threads)
    np.copyto(view.numpy(), host)
    with torch.cuda.stream(cs):
        out = view.to(dev, non_blocking=True)
    cs.synchronize()
    return out


@functools.lru_cache(maxsize=16)
def _rope_base_device(head_dim: int, nd: int, n_ids: int, device: str):
    c, s_ = _rope_base_table(head_dim, nd, n_ids)
    return torch.from_numpy(c).to(device).contiguous(), torch.from_numpy(s_).to(device).contiguous()


@functools.lru_cache(maxsize=16)
def _rope_base_interleaved(head_dim: int, nd: int, n_ids: int, device: str):
    """fp32 [n_ids + 1, F, 2] = (cos, sin) of every (position id, frequency), row n_ids = (1, 0): the table the to_qkv kernel gathers
    its rotary factors from (same values as _rope_base_table, i.e. as the reference's freqs_cis after the fp32 cast)."""
    c, s_ = _rope_base_table(head_dim, nd, n_ids)
    t = np.empty((n_ids + 1, c.shape[1], 2), dtype=np.float32)
    t[:n_ids, :, 0] = c
    t[:n_ids, :, 1] = s_
    t[n_ids, :, 0] = 1.0
    t[n_ids, :, 1] = 0.0
    return torch.from_numpy(t).to(device).contiguous()


def _rope_clip_table(grid: Tuple[int, ...], k: int, head_dim: int) -> np.ndarray:
    """fp32 [K+P, 64] = cos[32] | sin[32] for one clip; pairs beyond 3*F keep (1, 0) (rope.py:24 leaves them alone).
    Row ids: latent i -> (i,i,i); patch (t,h,w) -> (t,h,w) + K (rope.py:59-67); column f*nd + axis (interleaved)."""
    nd = len(grid)
    f = head_dim // (2 * nd)
    half = head_dim // 2
    n_ids = 1
    while n_ids < k + max(grid) + 1:
        n_ids *= 2
    n_ids = max(n_ids, 512)
    cos_t, sin_t = _rope_base_table(head_dim, nd, n_ids)
    p = int(np.prod(grid))
    ids = np.empty((k + p, nd), dtype=np.int64)
    ids[:k] = np.arange(k)[:, None]
    coords = np.indices(grid).reshape(nd, -1).T                     # raster order == torch.cartesian_prod
    ids[k:] = coords + k
    out = np.zeros((k + p, 2 * half), dtype=np.float32)
    out[:, :half] = 1.0
    # out[row, f*nd + axis] = table[ids[row, axis], f]
    out[:, : nd * f] = cos_t[ids].transpose(0, 2, 1).reshape(k + p, nd * f)
    out[:, half: half + nd * f] = sin_t[ids].transpose(0, 2, 1).reshape(k + p, nd * f)
    return out


@functools.lru_cache(maxsize=512)
def _clip_rope_ids(grid: Tuple[int, ...], k: int, n_ids: int) -> np.ndarray:
    """uint16 [K + P, 4] rotary position ids of one clip (rope.py:59-67): latent i -> (i, i, i), patch (t, h, w) -> (t, h, w) + K in raster
    order, slot 3 = the identity row of the base table (pairs 30, 31).  Read-only (cached)."""
    p = int(np.prod(grid))
    ids = np.empty((k + p, 4), dtype=np.uint16)
    ids[:, 3] = n_ids
    ids[:k, :3] = np.arange(k, dtype=np.uint16)[:, None]
    coords = np.indices(grid).reshape(len(grid), -1).T                      # raster order (t, h, w)
    ids[k:, :3] = np.minimum(coords + k, n_ids - 1).astype(np.uint16)
    ids.setflags(write=False)
    return ids


def _xcd_interleave(units) -> np.ndarray:
    """(sequence, first row) entries of the attention backward's 64-row blocks, ordered for the 8 XCDs: block b of a launch runs on XCD
    b % 8 under round-robin dispatch (a speed assumption only), and every block of a sequence streams that sequence's Q / dO (key
    blocks) or K / V (query blocks) tiles.  In sequence-major order one sequence's blocks were dealt over all 8 XCDs, every XCD's
    4 MB L2 saw every sequence, and the launch fetched 674 MB from HBM / MALL for ~100 MB of operands at the benchmark batch
    (profiles/r04_train_pmc.txt).  Sequences go to 8 lists (greedy by block count), entry i of the table comes from list i % 8; when the
    short lists run out the rest follows in list order (no padding entries: the kernels take the table as it is)."""
    if os.environ.get("TTV_BWD_XCD", "1") == "0":       # A/B: sequence-major order
        return np.asarray([e for u in units for e in u], dtype=np.int32).reshape(-1, 2)
    order = sorted(range(len(units)), key=lambda i: len(units[i]), reverse=True)
    lists, weight = [[] for _ in range(8)], [0] * 8
    for i in order:
        x = min(range(8), key=lambda j: weight[j])
        lists[x].extend(units[i])
        weight[x] += len(units[i])
    depth = min(len(l) for l in lists)
    out = [lists[x][k] for k in range(depth) for x in range(8)]
    for l in lists:
        out.extend(l[depth:])
    return np.asarray(out, dtype=np.int32).reshape(-1, 2)


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
        # rotary position ids per row (rope.py:59-67), four uint16 per row = two int32: (t | h << 16, w | identity << 16)
        n_ids = 512
        while n_ids < max(k + max(g) for g, k in zip(grids, self.token_counts)) + 1:
            n_ids *= 2
        self.n_rope_ids = n_ids
        # (per clip cached by (grid, K, table size): the loader's batches repeat a handful of clip shapes, and a step that needs three new
        # plans at ~5 clips is bound by this host code)
        ids = np.concatenate([_clip_rope_ids(grids[b], self.token_counts[b], n_ids) for b in range(B)], axis=0)
        rope_ids = ids.view(np.int32).reshape(-1)
        blocks64 = _xcd_interleave([[(b, r0) for r0 in range(0, cu[b + 1] - cu[b], 64)] for b in range(B)])
        row_seq = np.concatenate([np.full(cu[b + 1] - cu[b], b, dtype=np.int32) for b in range(B)])
        self.n_blocks64 = int(blocks64.shape[0])
        parts = [np.asarray(cu, dtype=np.int32), latent_rows, patch_rows, desc.reshape(-1), blocks64.reshape(-1), row_seq, rope_ids]
        offs, total = [], 0
        for p in parts:
            offs.append(total)
            total += (p.size + 3) // 4 * 4      # keep every table 16-byte aligned
        host = np.zeros(total, dtype=np.int32)
        for o, p in zip(offs, parts):
            host[o:o + p.size] = p
        self.int_tables = _upload(host, self.device)
        base = self.int_tables.data_ptr()
        if self.device.type == "cuda":
            # gather on the device from the (cached, device-resident) base table: no per-batch trigonometry, no big upload
            bc, bs = _rope_base_device(head_dim, len(self.patch), n_ids, str(self.device))
            self.rope_cs = torch.empty((self.total_rows, head_dim), dtype=torch.float32, device=self.device)
            rc = _lib.lib().ttv_rope_table_build(bc.data_ptr(), bs.data_ptr(), n_ids, bc.shape[1], base + 4 * offs[3], base + 4 * offs[0],
                                                 base + 4 * offs[5], self.rope_cs.data_ptr(), self.total_rows,
                                                 _lib.stream_ptr(self.device))
            _lib.check(rc, "ttv_rope_table_build")
            # the table is written by a kernel on the building stream: other streams wait for this event before their first use
            self.build_stream = torch.cuda.current_stream(self.device)
            self.ready = torch.cuda.Event()
            self.ready.record(self.build_stream)
        else:   # host tensors (CPU-side tests of the plan): same table, gathered with numpy
            rope = np.concatenate([_rope_clip_table(g, k, head_dim) for g, k in zip(grids, self.token_counts)], axis=0)
            self.rope_cs = torch.from_numpy(rope).to(self.device)

        self._base_fields = dict(
            n_clips=B, total_rows=self.total_rows, sum_tokens=self.sum_tokens, sum_patches=self.sum_patches,
            max_patches_per_clip=max(sizes),
            cu_seqlens=base + 4 * offs[0], latent_rows=base + 4 * offs[1], patch_rows=base + 4 * offs[2],
            clip_desc=base + 4 * offs[3], rope_cs=self.rope_cs.data_ptr(),
            blocks64=base + 4 * offs[4], row_seq=base + 4 * offs[5], n_blocks64=self.n_blocks64)
        # rotary factors by position id (ttv_batch.rope_ids / rope_base): the width-256 to_qkv kernel reads 8 bytes per row and gathers
        # from the cached base table instead of streaming the 256-byte fp32 row of rope_cs (TTV_ROPE_IDS=0: the table path, A/B)
        if self.device.type == "cuda" and len(self.patch) == 3 and os.environ.get("TTV_ROPE_IDS", "1") != "0":
            self._rope_base_cs = _rope_base_interleaved(head_dim, len(self.patch), n_ids, str(self.device))
            self._base_fields.update(rope_ids=base + 4 * offs[6], rope_base=self._rope_base_cs.data_ptr())
        self._offs = offs
        self._attn = {}
        self._attn_all_full = {}
        self._batch_structs = {}
        self.reader_streams = {}

    def use_on_current_stream(self) -> None:
        """Stream safety of the cached plan (see ForwardPipeline): wait for the build on foreign streams, remember the readers."""
        if self.device.type != "cuda":
            return
        cur = torch.cuda.current_stream(self.device)
        if cur != self.build_stream and cur.cuda_stream not in self.reader_streams:
            cur.wait_event(self.ready)
        self.reader_streams[cur.cuda_stream] = cur

    def retire(self) -> None:
        """Before the tables are released (cache eviction): the releasing stream waits for every stream that read them."""
        if self.device.type != "cuda":
            return
        cur = torch.cuda.current_stream(self.device)
        build = getattr(self, "build_stream", None)
        for st in self.reader_streams.values():
            if st != cur:
                cur.wait_stream(st)
            if build is not None and build != cur and st != build:      # the tables return to the build stream's pool (see _Tower._retire_pack)
                build.wait_stream(st)

    def attention_table(self, q_heads: int, kv_heads: int, split: Optional[bool] = None) -> torch.Tensor:
        """int32 [n,4] attention work table (sequence, first query row, q-head, mode) for this batch, XCD-aware.

        Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md: blocks b and b+8 share an XCD - a speed
        assumption only, never correctness).  All blocks of one (sequence, kv-head) unit re-read the same K/V, so units
        are distributed over 8 lists (greedy by block count) and the lists are interleaved: entry i goes to list i % 8.
        Shorter lists are padded with sequence = -1 entries (the kernel returns immediately).

        Half items (mode 1: 64 query rows, the key range split between the wave pairs of the block and merged in LDS) cost
        ~0.6 of a full item for half the work.  They pay when the grid alone cannot fill the part: with fewer items than the
        ATTN_SLOTS = 1024 blocks that are resident at once (batches below ~28 benchmark-size clips, e.g. the reference's
        6144-token training batches) the last third of every sequence's query blocks is issued as half items and the launch
        gets shorter (5 clips: 33 -> 27 us).  Larger grids keep full items: with a second batch in flight (pipeline.py) the
        tail of one launch is filled by the other chain and the extra work of half items only costs (-3 % measured), alone
        they are worth +1 % (TTV_ATTN_TAIL_DIV=8 halves the last eighth of every sequence's blocks on large grids: exactly
        the 128 items beyond 1024 slots at the benchmark batch, +1.5 % one batch at a time, -1 % with two in flight).  Within either regime the choice depends only on the sequence's own length, so a clip's bf16
        result does not depend on what it is packed with (the fp32 kernel computes the same way in both modes).
        `split`: None = that rule, False = never, True = every item (tests)."""
        if split is None and os.environ.get("TTV_ATTN_SPLIT") in ("0", "1"):      # diagnostics: A/B timing of the table kinds
            split = os.environ["TTV_ATTN_SPLIT"] == "1"
        key = (int(q_heads), int(kv_heads), split)
        t = self._attn.get(key)
        if t is None:
            rep = q_heads // kv_heads
            n_items = sum(-(-(self.cu_seqlens[b + 1] - self.cu_seqlens[b]) // QBLOCK) for b in range(len(self.grids))) * q_heads
            small_grid = n_items < ATTN_SLOTS
            units_full, units_half = [], []
            for b in range(len(self.grids)):
                s = self.cu_seqlens[b + 1] - self.cu_seqlens[b]
                nq = -(-s // QBLOCK)
                tail_div = 3 if small_grid else int(os.environ.get("TTV_ATTN_TAIL_DIV", "0"))
                first_half = 0 if split else (nq if (split is False or tail_div <= 0) else nq - nq // tail_div)
                for kvh in range(kv_heads):
                    heads = [kvh * rep + r for r in range(rep)]
                    units_full.append([(b, qb * QBLOCK, hd, 0) for qb in range(first_half) for hd in heads])
                    units_half.append([(b, q0, hd, 1) for qb in range(first_half, nq) for hd in heads
                                       for q0 in (qb * QBLOCK, qb * QBLOCK + 64) if q0 < s])
            # a (sequence, kv-head) unit keeps one XCD list for its full and its half items
            order = sorted(range(len(units_full)), key=lambda i: len(units_full[i]) + len(units_half[i]) / 2, reverse=True)
            lists, weight, halves = [[] for _ in range(8)], [0.0] * 8, [[] for _ in range(8)]
            for i in order:
                x = min(range(8), key=lambda j: weight[j])
                lists[x].extend(units_full[i])
                halves[x].extend(units_half[i])
                weight[x] += len(units_full[i]) + len(units_half[i]) / 2
            for x in range(8):
                lists[x].extend(halves[x])
            depth = max(len(l) for l in lists)
            table = np.full((depth, 8, 4), -1, dtype=np.int32)
            for x, l in enumerate(lists):
                if l:
                    table[: len(l), x, :] = np.asarray(l, dtype=np.int32)
            # drop trailing all-padding rows only (interior padding keeps the i % 8 alignment)
            flat = table.reshape(-1, 4)
            last = int(np.max(np.nonzero(flat[:, 0] >= 0)[0])) + 1
            t = _upload(np.ascontiguousarray(flat[:last]), self.device)
            self._attn[key] = t
            self._attn_all_full[t.data_ptr()] = not bool((flat[:last, 3] > 0).any())
        return t

    def attention_table_latent(self, q_heads: int, kv_heads: int) -> torch.Tensor:
        """int32 [n,4] attention work table like `attention_table`, restricted to the query blocks that hold LATENT tokens (rows
        [0, K_b) of every sequence; full items only).  The encoder reads its output from the latent rows alone (reference
        blocks.py:101-103), so its last layer needs attention outputs for these query rows only (`ttv_batch.qblocks_latent`): at the
        benchmark shape 128 entries instead of 1152.  Same XCD interleaving as the full table (a (sequence, kv-head) unit keeps one list)."""
        key = ("latent", int(q_heads), int(kv_heads))
        t = self._attn.get(key)
        if t is None:
            rep = q_heads // kv_heads
            units = []
            for b in range(len(self.grids)):
                nq = -(-int(self.token_counts[b]) // QBLOCK)
                for kvh in range(kv_heads):
                    units.append([(b, qb * QBLOCK, kvh * rep + r, 0) for qb in range(nq) for r in range(rep)])
            order = sorted(range(len(units)), key=lambda i: len(units[i]), reverse=True)
            lists, weight = [[] for _ in range(8)], [0] * 8
            for i in order:
                x = min(range(8), key=lambda j: weight[j])
                lists[x].extend(units[i])
                weight[x] += len(units[i])
            depth = max(len(l) for l in lists)
            table = np.full((depth, 8, 4), -1, dtype=np.int32)
            for x, l in enumerate(lists):
                if l:
                    table[: len(l), x, :] = np.asarray(l, dtype=np.int32)
            flat = table.reshape(-1, 4)
            last = int(np.max(np.nonzero(flat[:, 0] >= 0)[0])) + 1
            t = _upload(np.ascontiguousarray(flat[:last]), self.device)
            self._attn[key] = t
        return t

    def attention_table_patch(self, q_heads: int, kv_heads: int) -> Optional[torch.Tensor]:
        """int32 [n,4] attention work table like `attention_table` (full items only) WITHOUT the query blocks that hold latent rows only
        (block q of sequence b with (q + 1) * 128 <= K_b), or None when no sequence has such a block.  The decoder reads its output from
        the patch rows alone (reference blocks.py:171), so its last layer needs no attention output for those query rows
        (`ttv_batch.qblocks_patch`): at the benchmark shape 1024 entries instead of 1152.  Same XCD interleaving as the full table."""
        key = ("patch", int(q_heads), int(kv_heads))
        if key not in self._attn:
            rep = q_heads // kv_heads
            units, dropped = [], 0
            for b in range(len(self.grids)):
                s = self.cu_seqlens[b + 1] - self.cu_seqlens[b]
                nq = -(-s // QBLOCK)
                first = int(self.token_counts[b]) // QBLOCK          # blocks [0, first) hold latent rows only
                dropped += first
                for kvh in range(kv_heads):
                    units.append([(b, qb * QBLOCK, kvh * rep + r, 0) for qb in range(first, nq) for r in range(rep)])
            t = None
            if dropped and any(units):
                order = sorted(range(len(units)), key=lambda i: len(units[i]), reverse=True)
                lists, weight = [[] for _ in range(8)], [0] * 8
                for i in order:
                    x = min(range(8), key=lambda j: weight[j])
                    lists[x].extend(units[i])
                    weight[x] += len(units[i])
                depth = max(len(l) for l in lists)
                table = np.full((depth, 8, 4), -1, dtype=np.int32)
                for x, l in enumerate(lists):
                    if l:
                        table[: len(l), x, :] = np.asarray(l, dtype=np.int32)
                flat = table.reshape(-1, 4)
                last = int(np.max(np.nonzero(flat[:, 0] >= 0)[0])) + 1
                t = _upload(np.ascontiguousarray(flat[:last]), self.device)
            self._attn[key] = t
        return self._attn[key]

    def attention_table64(self, q_heads: int, kv_heads: int) -> torch.Tensor:
        """int32 [n,8] work table of ttv_attention64 (the 64-query-rows-per-wave kernel): one entry per workgroup =
        (sequence, kv-head, 4 x wave item, first packed row of the sequence, its length); a wave item is
        q-head | (first query row // 64) << 8, or -1 for an idle wave.

        All wave items of one (sequence, kv-head) unit - every 64-row slice of every q-head of the group - read the same K / V, so a
        unit's items are dealt four at a time to workgroups (the four waves of a workgroup share each K / V tile through LDS) and the
        unit's workgroups go to one of 8 lists (greedy by count); entry i of the flat table belongs to list i % 8 (blocks b, b + 8
        share an XCD under round-robin dispatch: L2 affinity only, never correctness).  Shorter lists are padded with sequence -1."""
        key = ("w64", int(q_heads), int(kv_heads))
        t = self._attn.get(key)
        if t is None:
            rep = q_heads // kv_heads
            units = []
            for b in range(len(self.grids)):
                s = self.cu_seqlens[b + 1] - self.cu_seqlens[b]
                n64 = -(-s // 64)
                for kvh in range(kv_heads):
                    items = [(kvh * rep + r) | (q << 8) for q in range(n64) for r in range(rep)]
                    items += [-1] * (-len(items) % 4)
                    units.append([(b, kvh, *items[i:i + 4], self.cu_seqlens[b], s) for i in range(0, len(items), 4)])
            order = sorted(range(len(units)), key=lambda i: len(units[i]), reverse=True)
            lists, weight = [[] for _ in range(8)], [0] * 8
            for i in order:
                x = min(range(8), key=lambda j: weight[j])
                lists[x].extend(units[i])
                weight[x] += len(units[i])
            depth = max(len(l) for l in lists)
            table = np.zeros((depth, 8, 8), dtype=np.int32)
            table[:, :, 0] = -1
            for x, l in enumerate(lists):
                if l:
                    table[: len(l), x, :] = np.asarray(l, dtype=np.int32)
            flat = table.reshape(-1, 8)
            last = int(np.max(np.nonzero(flat[:, 0] >= 0)[0])) + 1
            t = _upload(np.ascontiguousarray(flat[:last]), self.device)
            self._attn[key] = t
        return t

    def batch_for(self, q_heads: int, kv_heads: int) -> "_lib.Batch":
        """ttv_batch struct whose attention work table matches the tower's head counts (built once per head counts: every tower call of a
        step asks for it, and a ctypes struct of ~30 fields is 40 us of host time in a step that is launch-bound at small batches)."""
        skey = (int(q_heads), int(kv_heads)) + tuple(os.environ.get(k) for k in ("TTV_ATTN_SPLIT", "TTV_ATTN_PAIRED", "TTV_ATTN64", "TTV_ATTN_TAIL_DIV"))
        cached = self._batch_structs.get(skey)       # (the diagnostic switches that pick the tables are part of the key)
        if cached is not None:
            return cached
        t = self.attention_table(q_heads, kv_heads)
        # every (sequence, kv-head) unit contributes its entries in runs of `rep` q-heads per query block, full items first:
        # with an even rep, entries 2j and 2j+1 of an XCD list are two q-heads of one kv-head on the same query rows
        # (opt-in, TTV_ATTN_PAIRED=1: the 8-wave blocks halve the K/V tile traffic - +1.5 % with two batches in flight - but make
        # the grid coarser: the launch alone is 10 % slower)
        paired = 1 if (q_heads // kv_heads) % 2 == 0 and os.environ.get("TTV_ATTN_PAIRED", "0") == "1" else 0
        all_full = 1 if self._attn_all_full.get(t.data_ptr()) else 0
        # the 64-rows-per-wave kernel (ttv_attention64) for inference towers with pre-scaled q: OPT-IN (TTV_ATTN64=1).  Measured in
        # round 3 (profiles/r03_attn64_stamps.txt): per SIMD its key loop needs ~1 060 cycles per 32-query x
        # 64-key unit at two waves per SIMD against ~885 for the 32-rows-per-wave kernel at three, a block's prologue + epilogue
        # are 18 % of its life and 576 workgroups on 512 resident slots leave a tail: 77-80 us against 60-62 us at the benchmark
        # batch; at S = 9216 both kernels reach the same 0.37 of the MFMA peak (1 119 vs 1 120 us).
        t64 = self.attention_table64(q_heads, kv_heads) if (q_heads <= 255 and os.environ.get("TTV_ATTN64", "0") == "1") else None
        # the encoder's last layer on its latent rows only (ttv_batch.qblocks_latent; TTV_ENC_LATENT_LAST=0 in the library: A/B)
        tl = self.attention_table_latent(q_heads, kv_heads) if sum(int(k) for k in self.token_counts) > 0 else None
        # the decoder's last layer without the query blocks that hold latent rows only (ttv_batch.qblocks_patch; TTV_DEC_PATCH_LAST=0: A/B)
        # (only beside a table of full items: the last layer must run the kernel, and the item kind, the all-blocks forward runs)
        tp = self.attention_table_patch(q_heads, kv_heads) if all_full else None
        self._batch_structs[skey] = out = _lib.Batch(n_qblocks=int(t.shape[0]), qblocks=t.data_ptr(), qblocks_paired=paired, qblocks_all_full=all_full,
                          items64=t64.data_ptr() if t64 is not None else None, n_items64=int(t64.shape[0]) if t64 is not None else 0,
                          qblocks_latent=tl.data_ptr() if tl is not None else None, n_qblocks_latent=int(tl.shape[0]) if tl is not None else 0,
                          qblocks_patch=tp.data_ptr() if tp is not None else None, n_qblocks_patch=int(tp.shape[0]) if tp is not None else 0,
                          **self._base_fields)
        return out

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
            _plan_cache.pop(next(iter(_plan_cache))).retire()
        plan = BatchPlan(key[0], key[1], key[2], device)
        _plan_cache[key] = plan
    plan.use_on_current_stream()
    return plan


def host_ints(v) -> List[int]:
    """token_counts / grids as Python ints.  A device tensor costs one host sync (the reference syncs several times
    per tower for the same information); pass lists or CPU tensors to avoid it."""
    if isinstance(v, torch.Tensor):
        return v.detach().to("cpu").tolist()
    return [x.tolist() if hasattr(x, "tolist") else x for x in v]
