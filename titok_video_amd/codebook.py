"""Codebook usage statistics: the reference's CodebookLogger (train_utils/codebook_logging.py:5-37) without its per-step
device->host sync, plus the data-parallel reduction the reference does not have.

Reference semantics kept: a FIFO of the last `codebook_size` SAMPLES (one sample = the index tensor of one clip);
`get_scores()` = sum of the samples' bincounts -> usage percent and entropy (nats) of the normalised histogram, after
which the FIFO is cleared; `None` until the FIFO is full.

Differences: samples stay on the GPU (the reference calls `.cpu()` on every training step, train.py:114-115); the histogram
is built by the HIP kernel `ttv_codebook_histogram`; under torch.distributed the per-rank histograms are summed with ONE
all-reduce of `codebook_size` int64 counts (35 KB for tiny) so that every rank reports what a single logger seeing all
ranks' samples would report.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch import nn

from . import _lib


class CodebookLogger(nn.Module):
    def __init__(self, codebook_size: int, process_group=None, world_size: int = 1):
        super().__init__()
        self.codebook_size = int(codebook_size)
        # FIFO capacity in samples.  Single process: codebook_size, like the reference.  Data parallel: every rank keeps
        # its share so that the union over ranks is again the last `codebook_size` samples.
        self.capacity = max(1, self.codebook_size // max(1, int(world_size)))
        self.codebook_indices: List[torch.Tensor] = []
        self.process_group = process_group

    def forward(self, codes: Sequence[torch.Tensor]) -> None:
        for sample in codes:
            if len(self.codebook_indices) == self.capacity:
                self.codebook_indices.pop(0)
            self.codebook_indices.append(sample)

    def is_score_ready(self) -> bool:
        return len(self.codebook_indices) == self.capacity

    def histogram(self) -> torch.Tensor:
        """int64 [codebook_size] counts of everything currently in the FIFO (device of the samples)."""
        flat = torch.cat([s.reshape(-1) for s in self.codebook_indices]).to(torch.int32).contiguous()
        counts = torch.zeros(self.codebook_size, dtype=torch.int64, device=flat.device)
        if flat.is_cuda:
            rc = _lib.lib().ttv_codebook_histogram(flat.data_ptr(), flat.numel(), counts.data_ptr(), self.codebook_size,
                                                   _lib.stream_ptr(flat.device))
            _lib.check(rc, "ttv_codebook_histogram")
        else:   # host tensors (e.g. the gloo tests): same integer counting, nothing to accelerate
            counts += torch.bincount(flat.to(torch.int64), minlength=self.codebook_size)
        return counts

    @staticmethod
    def scores_from_counts(counts: torch.Tensor):
        """(usage percent, entropy in nats) exactly as codebook_logging.py:27-30 computes them from the histogram."""
        freq = counts.to(torch.float64).cpu()
        size = freq.numel()
        usage = float(torch.count_nonzero(freq)) / size * 100.0
        p = freq / freq.sum()
        nz = p[p > 0]
        entropy = float(-(nz * nz.log()).sum())
        return usage, entropy

    def get_scores(self, all_reduce: bool = True) -> Optional[dict]:
        """Scores once the FIFO is full, else None.  Under torch.distributed the decision is COLLECTIVE: ranks receive different
        numbers of clips per step (token-budget batching), so their FIFOs fill on different steps; every rank therefore enters the
        same two small all-reduces on every call - the number of ranks that are ready, then (only if all are) the histogram - and
        all ranks return scores on the same call.  A rank that is full keeps its newest `capacity` samples while it waits."""
        dist_on = all_reduce and torch.distributed.is_available() and torch.distributed.is_initialized()
        ready = self.is_score_ready()
        if dist_on:
            # the flag lives where the backend can reduce it: host memory for gloo, this rank's GPU for nccl (= RCCL) - also when the
            # FIFO is still empty and there is no sample to take the device from
            if torch.distributed.get_backend(self.process_group) == "gloo":
                dev = torch.device("cpu")
            else:
                dev = self.codebook_indices[0].device if self.codebook_indices else None
                if dev is None or dev.type != "cuda":
                    dev = torch.device("cuda", torch.cuda.current_device())
            flag = torch.tensor([1 if ready else 0], dtype=torch.int64, device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=self.process_group)
            ready = bool(int(flag.item()))
        if not ready:
            return None
        counts = self.histogram()
        if dist_on:
            if counts.is_cuda and torch.distributed.get_backend(self.process_group) == "gloo":
                host = counts.cpu()
                torch.distributed.all_reduce(host, op=torch.distributed.ReduceOp.SUM, group=self.process_group)
                counts = host
            else:
                torch.distributed.all_reduce(counts, op=torch.distributed.ReduceOp.SUM, group=self.process_group)
        usage, entropy = self.scores_from_counts(counts)
        self.codebook_indices = []
        return {"codebook/usage_percent": usage, "codebook/entropy": entropy}
