"""Shard loader that keeps up with the training step (BASELINE config #3).

The reference feeds its training loop from 3 loader worker processes (dataset/video_dataset.py:186-211: WebDataset shards ->
decode -> `_video_process` -> `_dynamic_batching`).  Round 2's stand-in decoded, uploaded and normalised clip by clip on a Python
thread of the training process and was the whole step: 45 ms per step against 3.5 ms of GPU work.  Here

  * WORKER PROCESSES (forked before the training process touches the GPU) read this rank's shards, keep the frames as the decoder
    hands them over - uint8 [T,H,W,3] - and run the reference's token-budget batching (data.dynamic_batches) on them; a batch
    crosses to the training process as shared-memory tensors (no pickling of pixel data);
  * one thread of the training process copies a batch into PINNED staging memory, uploads the uint8 frames (a quarter of the bytes
    of fp32 clips) on its own stream and normalises them ON THE GPU in one launch per clip (ttv_clip_from_u8: u8 / 127.5 - 1,
    dataset/video_dataset.py:116-119), records an event and hands the batch over;
  * the consumer makes the compute stream wait for that event and calls `record_stream` on every clip, so the allocator cannot hand
    a clip's memory back to the upload stream while the training step is still reading it.

Batches are dealt round-robin over the workers in a fixed order, so a (paths, rank, world, seed, workers) tuple always yields the
same batch sequence.
"""
from __future__ import annotations

import io
import json
import os
import queue
import tarfile
import threading
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch
import torch.multiprocessing as mp

from . import _lib
from .data import dynamic_batches


def raw_shard_samples(paths: Sequence[str], epochs: Optional[int] = 1) -> Iterator[Dict]:
    """Samples of the given shards, undecoded beyond what the decoder delivers: {'video': uint8 [T,H,W,3] CPU tensor, 'fps', '__key__'}.
    `video.shape[1:]` is what data.dynamic_batches sizes a clip by, so the dict carries a channel-first VIEW for it."""
    ep = 0
    while epochs is None or ep < epochs:
        for path in paths:
            with tarfile.open(path, "r") as tar:
                pending: Dict[str, Dict] = {}
                for m in tar:
                    key, ext = os.path.splitext(m.name)
                    rec = pending.setdefault(key, {})
                    rec[ext] = tar.extractfile(m).read()
                    if ".npy" in rec and ".json" in rec:
                        frames = torch.from_numpy(np.load(io.BytesIO(rec[".npy"]), allow_pickle=False))      # [T,H,W,3] uint8
                        yield {"video": frames.permute(3, 0, 1, 2), "fps": json.loads(rec[".json"])["fps"], "__key__": key}
                        del pending[key]
        ep += 1


def _worker_main(paths, patch, token_range, seq_len, seed, epochs, drop_last, out_q, release):
    torch.set_num_threads(1)
    try:
        for b in dynamic_batches(raw_shard_samples(paths, epochs), patch, token_range, seq_len, seed=seed, drop_last=drop_last):
            frames = [v.permute(1, 2, 3, 0).contiguous().share_memory_() for v in b["video"]]      # back to [T,H,W,3], in shared memory
            out_q.put({"frames": frames, "fps": b["fps"], "__key__": b["__key__"], "token_counts": b["token_counts"].tolist()})
    finally:
        out_q.put(None)
        release.wait()          # shared-memory tensors are handed over by file descriptor: stay alive until the consumer is done


class ShardBatchLoader:
    """Token-budget batches of this rank's shards (shard i -> rank i % world, a rank's shard j -> worker j % workers).

        loader = ShardBatchLoader(paths, rank, world, workers=2); loader.start()     # BEFORE the first GPU call of the process
        for batch in loader.batches(device, torch.bfloat16): ...                    # {'video': [C,T,H,W tensors], 'fps', '__key__', 'token_counts'}
    """

    def __init__(self, paths: Sequence[str], rank: int = 0, world_size: int = 1, patch=(4, 8, 8), token_range=(1, 128), seq_len: int = 6144,
                 seed: int = 0, workers: int = 2, epochs: Optional[int] = None, drop_last: bool = True, prefetch: int = 4):
        mine = [p for i, p in enumerate(sorted(paths)) if i % world_size == rank]
        self.workers = max(1, min(workers, len(mine)))
        self._args = [([p for j, p in enumerate(mine) if j % self.workers == w], tuple(patch), tuple(token_range), seq_len, seed + 1009 * w,
                       epochs, drop_last) for w in range(self.workers)]
        self.prefetch = prefetch
        self._procs, self._queues = [], []
        self._stop = threading.Event()

    def start(self) -> "ShardBatchLoader":
        if torch.cuda.is_initialized():
            raise RuntimeError("ShardBatchLoader.start(): fork the loader workers before the process touches the GPU")
        ctx = mp.get_context("fork")
        self._release = ctx.Event()
        for a in self._args:
            q = ctx.Queue(maxsize=self.prefetch)
            p = ctx.Process(target=_worker_main, args=(*a, q, self._release), daemon=True)
            p.start()
            self._procs.append(p)
            self._queues.append(q)
        return self

    def raw_batches(self) -> Iterator[Dict]:
        """The workers' batches in the fixed round-robin order (uint8 frames in shared memory); ends when every worker is done."""
        live = list(range(self.workers))
        while live:
            for w in list(live):
                b = self._queues[w].get()
                if b is None:
                    live.remove(w)
                    continue
                yield b

    def batches(self, device, dtype=torch.bfloat16, depth: int = 3) -> Iterator[Dict]:
        """Device batches.  A thread stages / uploads / normalises `depth` batches ahead on its own stream; the generator makes the
        CURRENT stream wait for a batch's event and marks its clips as used there (record_stream) before yielding it."""
        device = torch.device(device)
        lib = _lib.lib()
        code = _lib.dtype_code(dtype)
        out_q: "queue.Queue" = queue.Queue(maxsize=depth)

        def uploader():
            torch.cuda.set_device(device)
            up = torch.cuda.Stream(device=device)
            slots = [None] * (depth + 2)          # pinned staging buffers with the event of their last upload
            k = 0
            try:
                for b in self.raw_batches():
                    if self._stop.is_set():
                        break
                    n_bytes = sum(f.numel() for f in b["frames"])
                    slot = slots[k % len(slots)]
                    if slot is not None and slot[1] is not None:
                        slot[1].synchronize()      # the upload that last read this staging buffer has finished
                    if slot is None or slot[0].numel() < n_bytes:
                        slot = [torch.empty(max(n_bytes, 1 << 22), dtype=torch.uint8).pin_memory(), None]
                    off, views = 0, []
                    for f in b["frames"]:
                        v = slot[0][off:off + f.numel()].view(f.shape)
                        v.copy_(f)                 # shared memory -> pinned (a memcpy: releases the GIL)
                        views.append(v)
                        off += (f.numel() + 15) // 16 * 16
                    with torch.cuda.stream(up):
                        clips = []
                        for v in views:
                            t, h, w, _c = v.shape
                            d8 = v.to(device, non_blocking=True)
                            clip = torch.empty((3, t, h, w), dtype=dtype, device=device)
                            _lib.check(lib.ttv_clip_from_u8(d8.data_ptr(), t, h, w, clip.data_ptr(), code, up.cuda_stream), "ttv_clip_from_u8")
                            clips.append(clip)
                        ev = torch.cuda.Event()
                        ev.record(up)
                    slot[1] = ev
                    slots[k % len(slots)] = slot
                    k += 1
                    out_q.put(({"video": clips, "fps": b["fps"], "__key__": b["__key__"],
                                "token_counts": torch.tensor(b["token_counts"], dtype=torch.int32)}, ev))
            finally:
                out_q.put(None)

        th = threading.Thread(target=uploader, daemon=True)
        th.start()
        while True:
            item = out_q.get()
            if item is None:
                return
            batch, ev = item
            cur = torch.cuda.current_stream(device)
            cur.wait_event(ev)
            for c in batch["video"]:
                c.record_stream(cur)               # allocated on the upload stream, read on this one
            yield batch

    def close(self) -> None:
        self._stop.set()
        if getattr(self, "_release", None) is not None:
            self._release.set()
        for p in self._procs:
            p.join(timeout=2)
        for p in self._procs:
            if p.is_alive():
                p.terminate()
        for p in self._procs:
            p.join(timeout=5)
        self._procs, self._queues = [], []
