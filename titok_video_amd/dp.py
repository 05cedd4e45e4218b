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


class GradReducer:
    """Count-weighted gradient all-reduce OVERLAPPED with the backward pass (the exchange step of the reference's training step,
    train.py:75-83, as BASELINE.json's north_star states it: "RCCL all-reduce of grads ... overlapped with backward").

    The towers' hand-written backward writes every parameter gradient into ONE flat fp32 buffer per tower in which a layer's
    gradients are contiguous, and records an event per layer as soon as that layer's slice is final (ttv_tower_grads.
    layer_done_events; the backward visits the layers top down).  `reduce_slice` is called by the tower's autograd function right
    after the backward kernels have been ENQUEUED: a communication stream waits for the slice's event, multiplies the slice by
    this rank's clip count, all-reduces it in place (RCCL over xGMI: one message per layer slice, 3-30 MB - ring collectives are
    bound per link, so fewer / larger messages; no concatenation, no copy) and divides by the global count, while the compute
    stream goes on with the layers below and the next tower.  The result is sum_r(count_r * grad_r) / sum_r(count_r): the
    single-process mean over the union batch also when ranks hold different numbers of clips.

    The global clip count is exchanged at `begin_step` (one 8-byte all-reduce on the communication stream, long finished when the
    first slice arrives) and stays on the device: no host synchronisation anywhere between backward and the optimizer step.
    `finish()` makes the current stream wait for the communication stream (call it before clipping / the optimizer).

    backend "gloo" (CPU-side collective; the tests' stand-in for RCCL, both ranks on one GPU): same arithmetic, the slice is
    staged through the host - which synchronises, so nothing overlaps there; the results are bit-identical to RCCL's order of
    operations (multiply, sum over ranks, divide) and to `allreduce_mean_by_count`."""

    def __init__(self, device, group=None, simulate: bool = False):
        """simulate=True (tests on one GPU, no process group): the communication-stream path runs exactly as under RCCL - events,
        stream hand-over, deferred gradients - with the collective itself left out (world size 1: it would be the identity)."""
        self.device = torch.device(device)
        self.group = group
        self.rank, self.world = world()
        self.simulate = bool(simulate) and self.world == 1
        self.gloo = self.world > 1 and dist.get_backend(group) == "gloo"
        self.comm = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._local = None      # fp32 device scalar: this rank's clip count
        self._total = None      # fp32 device scalar: the global clip count
        self.bytes_reduced = 0
        self.slices = 0
        self._deferred = []     # (parameter, gradient produced on the communication stream): assigned in finish()
        self.reduced_ids = set()   # id() of every parameter whose gradient went through this reducer in the current step

    @property
    def on_comm_stream(self) -> bool:
        """True when slices are reduced asynchronously on the communication stream (RCCL, or the one-GPU simulation)."""
        return self.comm is not None and ((self.world > 1 and not self.gloo) or self.simulate)

    def attach(self, *towers) -> None:
        for t in towers:
            t._grad_reducer = self

    def detach(self, *towers) -> None:
        for t in towers:
            t.__dict__.pop("_grad_reducer", None)

    def begin_step(self, local_count: int) -> None:
        self.bytes_reduced, self.slices = 0, 0
        self._deferred, self.reduced_ids = [], set()
        if self.simulate:
            self.comm.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm):
                self._local = torch.full((), float(local_count), dtype=torch.float32, device=self.device)
                self._total = self._local.clone()
            return
        if self.world == 1:
            return
        if self.gloo:
            cnt = torch.tensor([float(local_count)], dtype=torch.float64)
            dist.all_reduce(cnt, group=self.group)
            self._local = torch.tensor(float(local_count), dtype=torch.float32, device=self.device)
            self._total = torch.tensor(float(cnt.item()), dtype=torch.float32, device=self.device)
            return
        self.comm.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.comm):
            self._local = torch.full((), float(local_count), dtype=torch.float32, device=self.device)
            tot = self._local.clone()
            dist.all_reduce(tot, group=self.group)
            self._total = tot

    def reduce_slice(self, flat: torch.Tensor, lo: int, hi: int, ready: "torch.cuda.Event") -> None:
        """In place on flat[lo:hi] (fp32), after `ready`, on the communication stream."""
        if (self.world == 1 and not self.simulate) or hi <= lo:
            return
        sl = flat[lo:hi]
        self.bytes_reduced += 4 * (hi - lo)
        self.slices += 1
        if self.gloo:
            ready.synchronize()
            host = (sl * self._local).cpu()
            dist.all_reduce(host, group=self.group)
            sl.copy_(host.to(sl.device))
            sl.div_(self._total)
            return
        self.comm.wait_event(ready)
        with torch.cuda.stream(self.comm):
            sl.mul_(self._local)
            if not self.simulate:
                dist.all_reduce(sl, group=self.group)
            sl.div_(self._total)
        flat.record_stream(self.comm)

    def stream(self):
        """Context in which a tower turns its reduced flat buffer into per-parameter gradients (casts / permutations run behind
        the reductions on the communication stream instead of making the compute stream wait for them)."""
        import contextlib
        if not self.on_comm_stream:
            return contextlib.nullcontext()
        return torch.cuda.stream(self.comm)

    def defer(self, pairs) -> None:
        """(parameter, gradient) pairs whose gradient tensors were produced on the communication stream.  They are NOT handed to
        autograd (its AccumulateGrad runs on the compute stream without waiting for this one: any real accumulation - p.grad already
        set, a tower used twice in one backward, a tensor hook - would read the slice while the all-reduce is still writing it);
        `finish()` assigns / accumulates them behind the stream join."""
        self._deferred.extend(pairs)

    def finish(self) -> None:
        """Join the communication stream into the current one, then deliver the deferred gradients (p.grad = g, or p.grad += g when a
        gradient is already there - micro-batch accumulation, zero_grad(set_to_none=False))."""
        if self.on_comm_stream:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.comm)
            for p, g in self._deferred:
                g.record_stream(cur)
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
        self._deferred = []

    def reduce_rest(self, params, local_count: int) -> None:
        """Gradients of trainable parameters that did NOT go through the attached towers (e.g. an L2 quantiser's codebook, a future
        pre-/post-quantiser linear): reduced after the backward in flat buckets, same count-weighted mean.  Call after finish()."""
        rest = [p.grad for p in params if p.grad is not None and id(p) not in self.reduced_ids]
        if rest and self.world > 1:
            allreduce_mean_by_count(rest, local_count, group=self.group)


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
        # element-wise: (count * g) summed over ranks, then a true fp32 division by the total - written with tensor operands so that
        # it is the same arithmetic as GradReducer's device-side scalars (torch turns `x / python_float` into `x * (1 / float)`)
        flat = torch.cat([g.reshape(-1).to(torch.float32) for g in bucket])
        flat = flat * torch.tensor(float(local_count), dtype=torch.float32, device=flat.device)
        if flat.is_cuda and dist.get_backend(group) == "gloo":     # CPU-side collective (tests); RCCL reduces in place on the GPU
            host = flat.cpu()
            dist.all_reduce(host, group=group)
            flat = host.to(flat.device)
        else:
            dist.all_reduce(flat, group=group)
        flat = flat / torch.tensor(total, dtype=torch.float32, device=flat.device)
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
