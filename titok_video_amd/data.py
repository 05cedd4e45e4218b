"""Synthetic clip stream + token-budget dynamic batching.

The reference's loaders decode real videos (decord, WebDataset shards: out of scope).  What the hot path sees from them is
a stream of batch dicts built by `_dynamic_batching` (reference dataset/video_dataset.py:130-172): clips of varying
(T,H,W) are appended, each with a latent-token count drawn from `token_range`, until adding the next clip would push the
packed sequence length  sum(grid_size + token_count)  past the budget (`train_seq_len` 6144 / `eval_seq_len` 4096,
configs/tiny.yaml:65-66); the batch is then emitted as
    {'video': [C,T,H,W tensors], 'fps': [...], '__key__': [...], 'token_counts': int32 tensor [B]}.
This module produces the same dicts from seeded synthetic clips, with rank-disjoint sharding for data parallelism
(the reference has no node split - SURVEY.md R4).
"""
from __future__ import annotations

import math
import random
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import torch


def sample_clip_shape(rng: random.Random, min_grid: Sequence[int], max_grid: Sequence[int], patch: Sequence[int],
                      max_aspect_ratio: float = 2.0) -> Tuple[int, int, int]:
    """(T,H,W) on the patch lattice inside [min_grid, max_grid] with max(H,W)/min(H,W) <= max_aspect_ratio
    (the sampling ranges of configs/tiny.yaml:56-62)."""
    for _ in range(1000):
        shape = tuple(rng.randrange(lo // p, hi // p + 1) * p for lo, hi, p in zip(min_grid, max_grid, patch))
        if max(shape[1], shape[2]) <= max_aspect_ratio * min(shape[1], shape[2]):
            return shape
    raise RuntimeError("no clip shape satisfies the sampling constraints")


class SyntheticClipStream:
    """Infinite (or `length`-bounded) stream of {'video', 'fps', '__key__'} samples.  Sample i belongs to rank i % world_size."""

    def __init__(self, min_grid=(8, 128, 128), max_grid=(16, 168, 168), patch=(4, 8, 8), fps_range=(3, 5), max_aspect_ratio=2.0,
                 dtype=torch.bfloat16, device="cpu", seed: int = 0, rank: int = 0, world_size: int = 1, length: Optional[int] = None):
        self.min_grid, self.max_grid, self.patch = tuple(min_grid), tuple(max_grid), tuple(patch)
        self.fps_range, self.max_aspect_ratio = tuple(fps_range), max_aspect_ratio
        self.dtype, self.device, self.seed, self.rank, self.world_size, self.length = dtype, device, seed, rank, world_size, length

    def __iter__(self) -> Iterator[Dict]:
        i = self.rank
        while self.length is None or i < self.length:
            rng = random.Random((self.seed << 20) + i)               # per-sample seed: any rank can regenerate any sample
            shape = sample_clip_shape(rng, self.min_grid, self.max_grid, self.patch, self.max_aspect_ratio)
            g = torch.Generator(device="cpu").manual_seed((self.seed << 20) + i)
            video = torch.rand((3, *shape), generator=g, dtype=torch.float32) * 2.0 - 1.0     # [-1, 1] (video_dataset.py:118-119)
            yield {"video": video.to(device=self.device, dtype=self.dtype), "fps": rng.uniform(*self.fps_range), "__key__": f"synthetic_{i:08d}"}
            i += self.world_size


def dynamic_batches(samples, patch: Sequence[int], token_range: Sequence[int], max_seq_len: int, seed: int = 0,
                    max_grid: Optional[Sequence[int]] = None, device=None, drop_last: bool = False) -> Iterator[Dict]:
    """Token-budget batching with the reference's policy (video_dataset.py:130-172): never exceed `max_seq_len` packed rows.
    drop_last: the reference never emits the clips left over when the stream ends (its generator only yields when the NEXT clip
    would overflow the budget); True reproduces that, False (default, inference / tests) also emits the trailing partial batch."""
    if max_grid is not None and math.prod(x // y for x, y in zip(max_grid, patch)) + token_range[1] > max_seq_len:
        raise ValueError("max_grid/patch + token_range[1] must fit in max_seq_len")
    rng = random.Random(seed)
    chunk: List[Dict] = []
    counts: List[int] = []
    cur = 0
    for sample in samples:
        grid = math.prod(x // y for x, y in zip(sample["video"].shape[1:], patch))
        k = rng.randrange(token_range[0], token_range[1] + 1)
        if cur + grid + k > max_seq_len and chunk:
            yield _collate(chunk, counts, device)
            chunk, counts, cur = [], [], 0
        cur += grid + k
        chunk.append(sample)
        counts.append(k)
    if chunk and not drop_last:
        yield _collate(chunk, counts, device)


_control_groups: Dict = {}       # rank tuple of the data group (None = the world) -> gloo group


def _group_key(process_group):
    import torch.distributed as dist
    return None if process_group is None else tuple(dist.get_process_group_ranks(process_group))


def setup_control_group(process_group=None, force: bool = False):
    """COLLECTIVE over the WORLD: create (once) the CPU-side gloo group that carries the 8-byte control collectives of
    `process_group`.  `dist.new_group` must be entered by every rank of the default group, also by ranks outside `process_group`
    and by ranks that never iterate `equal_steps` (an evaluation-only rank) - so a program with sub-groups or such ranks calls this
    on every rank right after `init_process_group`.  When every rank of the world trains over the default group, `equal_steps`
    calls it lazily at its first step, which is then the same collective.  Cached by the group's RANK TUPLE (not `id()`, which a
    new group can reuse after the old one is collected); `drop_control_groups()` forgets them (call it before
    `destroy_process_group`)."""
    import torch.distributed as dist
    if dist.get_backend(process_group) == "gloo" and not force:
        return process_group
    key = _group_key(process_group)
    g = _control_groups.get(key)
    if g is None:
        g = _control_groups[key] = dist.new_group(ranks=list(key) if key is not None else None, backend="gloo")
    return g


def drop_control_groups() -> None:
    _control_groups.clear()


def _control_group(process_group=None, force: bool = False):
    """Under RCCL a flag all-reduce lives on the GPU and reading it back (`flag.item()`) drains everything the rank has queued - one
    host synchronisation per training step.  Control decisions go through gloo instead (`setup_control_group`): the host blocks for
    the other ranks' hosts only, the GPU queues stay full.  gloo groups are returned as they are."""
    return setup_control_group(process_group, force)


def equal_steps(batches: Iterator[Dict], process_group=None, _force_control_group: bool = False) -> Iterator[Dict]:
    """Data-parallel training loop guard: yield this rank's batches only while EVERY rank still has one.  Token-budget batching
    gives the ranks different numbers of batches for the same number of clips; the per-step gradient all-reduce must be entered
    by all ranks or by none, so the epoch ends (collectively) when the first rank runs out - one 8-byte all-reduce per step, on a
    CPU-side control group (no GPU synchronisation: `_control_group`)."""
    import torch.distributed as dist
    it = iter(batches)
    on = dist.is_available() and dist.is_initialized()
    ctl = _control_group(process_group, _force_control_group) if on else None
    while True:
        nxt = next(it, None)
        if on:
            flag = torch.tensor([0 if nxt is None else 1], dtype=torch.int64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctl)
            if int(flag.item()) == 0:
                return
        elif nxt is None:
            return
        yield nxt


def _collate(chunk: List[Dict], counts: List[int], device) -> Dict:
    out = {k: [c[k] for c in chunk] for k in chunk[0].keys()}
    out["token_counts"] = torch.tensor(counts, dtype=torch.int32, device=device if device is not None else "cpu")
    return out
