"""Several independent batches in flight on separate HIP streams.

Clips are independent in the tokenizer (block-diagonal attention over `cu_seqlens`, per-row norms, per-element FSQ: SURVEY.md
section 8e), so consecutive inference batches have no dependency on each other.  A single forward is a chain of ~27 dependent
launches, and several of them cannot fill the part on their own at the reference's batch sizes (the layer-tail kernel runs on
192 of 256 CUs, every launch ends in a partly filled round of resident blocks, launches are separated by a few microseconds):
with a second batch in flight on another stream the hardware fills those holes with the other chain's blocks.  Measured at the
benchmark batch (32 clips of 16x128x128, tiny): 20.9 k clips/s one batch at a time, 26.0 k clips/s with two in flight
(tools/two_streams.py).  Nothing about a batch's arithmetic changes: results are bit-identical to the sequential call.

    pipe = ForwardPipeline(model, depth=2)
    for clips, counts in loader:
        ticket = pipe.submit(clips, counts)      # returns at once; the forward is enqueued on one of `depth` streams
        ...
        recon, out = pipe.result(ticket)         # makes the CURRENT stream wait for that batch (no host sync)

Not for the training step: there every step depends on the weights the previous one produced.

Shared, lazily built device state is ordered across the streams by the objects themselves: a tower's packed weights and a cached
batch plan record an event on the stream that builds them and every other stream waits for it before its first use
(`_WeightPack.use_on_current_stream`, `BatchPlan.use_on_current_stream`); when one of them is replaced or evicted, the releasing
stream first waits for every stream that read it.  Input tensors are `record_stream`-ed on the side stream, so a caller may drop
its references as soon as `submit` returns.  `drain()` is still the way to make the current stream wait for everything in flight.
"""
from __future__ import annotations

import os
from typing import Any, Callable, List, Optional, Sequence, Tuple

import torch


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v)


class ForwardPipeline:
    def __init__(self, model: torch.nn.Module, depth: int = 2, device: Optional[torch.device] = None,
                 fn: Optional[Callable[..., Any]] = None):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.model = model
        self.fn = fn if fn is not None else model.__call__
        self.device = device if device is not None else next(model.parameters()).device
        if torch.device(self.device).type != "cuda":
            raise RuntimeError("ForwardPipeline needs a GPU model (there is no CPU path)")
        # TTV_PIPE_PRIO=1 (A/B): the streams get different queue priorities (the first one high), so that one chain is served first and
        # the other fills what it leaves idle, instead of two equal queues splitting every CU
        prio = os.environ.get("TTV_PIPE_PRIO", "0") == "1"
        self.streams = [torch.cuda.Stream(device=self.device, priority=(-1 if (prio and i == 0) else 0)) for i in range(depth)]
        self._n = 0

    def submit(self, *args, **kwargs) -> Tuple[Any, torch.cuda.Event]:
        """Enqueue fn(*args) on the next stream of the ring (after everything already queued on the current stream, so inputs
        produced there are complete).  Returns a ticket for `result`."""
        s = self.streams[self._n % len(self.streams)]
        self._n += 1
        s.wait_stream(torch.cuda.current_stream(self.device))
        for t in _tensors((args, kwargs)):
            if t.is_cuda:
                t.record_stream(s)       # the caller may drop its reference (`for clips in loader`) while the side stream still reads
        with torch.cuda.stream(s), torch.no_grad():
            out = self.fn(*args, **kwargs)
        done = torch.cuda.Event()
        done.record(s)
        return out, done

    def result(self, ticket):
        """Outputs of a submitted batch, safe to consume on the current stream (stream-side wait only)."""
        out, done = ticket
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)
        for t in _tensors(out):
            t.record_stream(cur)        # allocated on the side stream: tell the allocator who else uses it
        return out

    def drain(self) -> None:
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)
