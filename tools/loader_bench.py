#!/usr/bin/env python3
"""Where the config #3 step time goes: loader workers alone, upload + GPU normalisation alone, the training step on a resident batch.
GPU box only.    python tools/loader_bench.py"""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd.loader import ShardBatchLoader  # noqa: E402
from titok_video_amd.shards import write_synthetic_shards  # noqa: E402

d = os.path.join(tempfile.gettempdir(), "ttv_loader_bench")
paths = write_synthetic_shards(d, 4, 64, seed=11)
N = int(os.environ.get("N", "60"))
W = int(os.environ.get("WORKERS", "2"))
ld = ShardBatchLoader(paths, 0, 1, seed=100, workers=W).start()
it = ld.raw_batches()
next(it)
t0 = time.perf_counter()
n_clips = 0
for _ in range(N):
    n_clips += len(next(it)["frames"])
t = time.perf_counter() - t0
print(f"workers alone ({W} processes): {1e3 * t / N:.2f} ms per batch, {n_clips / N:.1f} clips per batch", flush=True)
ld.close()

ld = ShardBatchLoader(paths, 0, 1, seed=100, workers=W).start()
dev = torch.device("cuda:0")
it = ld.batches(dev, torch.bfloat16)
b = next(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    b = next(it)
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"workers + pinned staging + upload + GPU normalisation: {1e3 * t / N:.2f} ms per batch", flush=True)
ld.close()
