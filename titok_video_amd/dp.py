"""Single-node data parallelism for the tokenizer path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for the tests).

The reference has no distributed code at all; what must be preserved is single-process semantics:
  * encode/decode throughput path: clips are independent (block-diagonal attention, per-row norms, per-element FSQ), so
    clips are sharded over ranks and NO data-path collective is needed - `shard_clips`;
  * codebook statistics: one all-reduce(sum) of the int64 usage histogram (codebook.CodebookLogger.get_scores);
  * training: gradients all-reduced as SUM over ranks of (per-rank sum over clips) / (global clip count), which equals the
    single-process mean over the union batch even when ranks hold different clip counts - `allreduce_mean_by_count`
    (equal-weight averaging would be wrong for ragged per-rank batches).
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_clips(n_items: int, rank: int, world_size: int) -> List[int]:
    """Indices of the clips rank `rank` owns: round-robin, rank-disjoint, union = all clips."""
    return list(range(rank, n_items, world_size))


def gather_variable(t: torch.Tensor, group=None) -> List[torch.Tensor]:
    """all_gather of 1-D tensors whose lengths differ per rank (token indices of ragged shards)."""
    rank, ws = world()
    if ws == 1:
        return [t]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n, group=group)
    m = int(max(int(s) for s in sizes))
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    outs = [torch.zeros_like(pad) for _ in range(ws)]
    dist.all_gather(outs, pad, group=group)
    return [o[: int(s)] for o, s in zip(outs, sizes)]


def allreduce_mean_by_count(grads: Iterable[torch.Tensor], local_count: int, group=None, bucket_bytes: int = 64 << 20) -> int:
    """In-place: grads <- sum_over_ranks(local_count * grads) / sum_over_ranks(local_count).

    `grads` hold each rank's gradient of its LOCAL mean loss over `local_count` clips.  Buckets are flattened per dtype and
    reduced in one collective each (xGMI rings are per-link bound: fewer, larger messages).  Returns the global clip count."""
    rank, ws = world()
    grads = [g for g in grads if g is not None]
    if ws == 1:
        return int(local_count)
    dev = grads[0].device if grads else torch.device("cpu")
    cnt = torch.tensor([float(local_count)], dtype=torch.float64, device="cpu" if dist.get_backend(group) == "gloo" else dev)
    dist.all_reduce(cnt, group=group)
    total = float(cnt.item())
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1).to(torch.float32) for g in bucket]) * float(local_count)
        if flat.is_cuda and dist.get_backend(group) == "gloo":     # CPU-side collective (tests); RCCL reduces in place on the GPU
            host = flat.cpu()
            dist.all_reduce(host, group=group)
            flat = host.to(flat.device)
        else:
            dist.all_reduce(flat, group=group)
        flat /= total
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g).to(g.dtype))
            off += n
        bucket, size = [], 0

    for g in grads:
        bucket.append(g)
        size += g.numel() * 4
        if size >= bucket_bytes:
            flush()
    flush()
    return int(round(total))
