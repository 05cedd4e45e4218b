"""Synthetic WebDataset-style shards: a `tarfile` writer / reader standing in for the reference's data path around the hot path
(`webdataset`, `decord` and ffmpeg are absent here and out of scope).

Reference format (dataset/convert_to_wds.py:34-38): one tar member pair per sample, `{__key__}.mp4` holding an encoded video, read back
by `wds.tarfile_to_samples` and decoded / normalised by `_video_process` (dataset/video_dataset.py:38-127) into `[C,T,H,W]` in
[-1, 1] in the model dtype, then batched by `_dynamic_batching` (:130-172).  Here a member is `{__key__}.npy`: the DECODED frames as
uint8 `[T,H,W,3]` (what decord hands the reference after decoding), plus `{__key__}.json` with the fps; the reader normalises
`x / 127.5 - 1` exactly like the reference's `v2.ToDtype(scale=True)` + `Normalize(0.5, 0.5)` (video_dataset.py:116-119) and yields the
sample dicts `data.dynamic_batches` consumes.  Shards are dealt to ranks whole (shard i -> rank i % world_size): rank-disjoint,
which the reference's loader does not do (`split_by_worker` only).
"""
from __future__ import annotations

import io
import json
import os
import random
import tarfile
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from .data import sample_clip_shape


def write_synthetic_shards(out_dir: str, n_shards: int, clips_per_shard: int, min_grid=(8, 128, 128), max_grid=(16, 168, 168),
                           patch=(4, 8, 8), fps_range=(3, 5), max_aspect_ratio: float = 2.0, seed: int = 0) -> List[str]:
    """`n_shards` tar files `shard-%05d.tar` of `clips_per_shard` seeded uint8 clips each.  Returns the paths."""
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for s in range(n_shards):
        path = os.path.join(out_dir, f"shard-{s:05d}.tar")
        with tarfile.open(path, "w") as tar:
            for j in range(clips_per_shard):
                i = s * clips_per_shard + j
                rng = random.Random((seed << 20) + i)
                t, h, w = sample_clip_shape(rng, min_grid, max_grid, patch, max_aspect_ratio)
                g = np.random.default_rng((seed << 20) + i)
                frames = g.integers(0, 256, size=(t, h, w, 3), dtype=np.uint8)
                key = f"synthetic_{i:08d}"
                for name, payload in ((key + ".npy", _npy_bytes(frames)), (key + ".json", json.dumps({"fps": rng.uniform(*fps_range)}).encode())):
                    info = tarfile.TarInfo(name)
                    info.size = len(payload)
                    tar.addfile(info, io.BytesIO(payload))
        paths.append(path)
    return paths


def _npy_bytes(a: np.ndarray) -> bytes:
    buf = io.BytesIO()
    np.save(buf, a, allow_pickle=False)
    return buf.getvalue()


def shard_samples(paths: Sequence[str], rank: int = 0, world_size: int = 1, dtype=torch.bfloat16, device="cpu",
                  epochs: Optional[int] = 1) -> Iterator[Dict]:
    """Samples of the shards this rank owns (path i -> rank i % world_size), in order; `epochs=None` repeats forever (the reference's
    training loader resamples shards indefinitely, video_dataset.py:187).  Yields {'video': [3,T,H,W] in [-1,1], 'fps', '__key__'}."""
    mine = [p for i, p in enumerate(sorted(paths)) if i % world_size == rank]
    ep = 0
    while epochs is None or ep < epochs:
        for path in mine:
            with tarfile.open(path, "r") as tar:
                pending: Dict[str, Dict] = {}
                for m in tar:
                    key, ext = os.path.splitext(m.name)
                    data = tar.extractfile(m).read()
                    rec = pending.setdefault(key, {})
                    rec[ext] = data
                    if ".npy" in rec and ".json" in rec:
                        frames = np.load(io.BytesIO(rec[".npy"]), allow_pickle=False)                 # [T,H,W,3] uint8
                        # uint8 goes to the device first (a quarter of the bytes, and the normalisation runs there)
                        u8 = torch.from_numpy(frames).to(device)
                        video = (u8.permute(3, 0, 1, 2).to(torch.float32) / 127.5 - 1.0).to(dtype).contiguous()
                        yield {"video": video, "fps": json.loads(rec[".json"])["fps"], "__key__": key}
                        del pending[key]
        ep += 1
