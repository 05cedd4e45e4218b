"""Shard loader that keeps up with the training step (BASELINE config #3).

The reference feeds its training loop from 3 loader worker processes (dataset/video_dataset.py:186-211: WebDataset shards ->
decode -> `_video_process` -> `_dynamic_batching`).  Round 2's stand-in decoded, uploaded and normalised clip by clip on a Python
thread of the training process and was the whole step: 45 ms per step against 3.5 ms of GPU work.  Here

  * WORKER PROCESSES (forked before the training process touches the GPU) read this rank's shards, keep the frames as the decoder
    hands them over - uint8 [T,H,W,3] - and run the reference's token-budget batching (data.dynamic_batches) on them.  A batch's
    frames are written into one slot of a ring of shared-memory buffers allocated ONCE before the fork; only the slot number and
    the clip shapes cross the queue.  (Fresh shared-memory tensors per batch cost the consumer 20-30 ms per batch in first-touch
    page faults of the new mappings - measured, tools/loader_bench.py.)
  * the training process registers those rings as PINNED host memory (hipHostRegister) after its first GPU call; one thread uploads
    a batch's uint8 frames straight from the ring (a quarter of the bytes of fp32 clips, no staging copy) on its own stream and
    normalises them ON THE GPU in one launch per clip (ttv_clip_from_u8: u8 / 127.5 - 1, dataset/video_dataset.py:116-119),
    records an event, hands the batch over and returns the slot to its worker once the event has completed;
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
import traceback
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


class LoaderError(RuntimeError):
    """A loader worker process or the upload thread failed; the message carries the original traceback.  Raised in the CONSUMER
    (`raw_batches()` / `batches()`), never swallowed into a normal end-of-data."""


class _Failure:
    """Queue sentinel of a failed producer (picklable: it crosses the process boundary)."""

    def __init__(self, where: str, text: str):
        self.where, self.text = where, text


def _worker_main(paths, patch, token_range, seq_len, seed, epochs, drop_last, ring, free_q, out_q, release):
    torch.set_num_threads(1)
    try:
        for b in dynamic_batches(raw_shard_samples(paths, epochs), patch, token_range, seq_len, seed=seed, drop_last=drop_last):
            slot = free_q.get()                    # flow control: a slot the consumer has finished uploading from
            buf, off, clips = ring[slot], 0, []
            for v in b["video"]:                   # channel-first view of [T,H,W,3] frames: write them back as the decoder's layout
                t, h, w = v.shape[1:]
                n = 3 * t * h * w
                buf[off:off + n].view(t, h, w, 3).copy_(v.permute(1, 2, 3, 0))
                clips.append((off, t, h, w))
                off += (n + 15) // 16 * 16
            out_q.put({"slot": slot, "clips": clips, "fps": b["fps"], "__key__": b["__key__"], "token_counts": b["token_counts"].tolist()})
        out_q.put(None)                            # end of data: only a worker that finished its shards says so
    except BaseException:                          # corrupt shard, a clip that does not fit its ring slot, ...: the consumer re-raises
        out_q.put(_Failure(f"loader worker (shards {list(paths)})", traceback.format_exc()))
    finally:
        release.wait()


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
        self.prefetch = max(2, prefetch)
        # a batch holds at most seq_len patch rows of prod(patch) * 3 bytes each (+ 16-byte alignment per clip)
        self.slot_bytes = int(seq_len) * int(np.prod(patch)) * 3 + 4096
        self._procs, self._queues, self._free, self._rings = [], [], [], []
        self._registered = False
        self._pinned = False
        self._pinned_rings: List[torch.Tensor] = []
        self._stop = threading.Event()

    def start(self) -> "ShardBatchLoader":
        if torch.cuda.is_initialized():
            raise RuntimeError("ShardBatchLoader.start(): fork the loader workers before the process touches the GPU")
        ctx = mp.get_context("fork")
        self._release = ctx.Event()
        for a in self._args:
            ring = torch.zeros(self.prefetch, self.slot_bytes, dtype=torch.uint8).share_memory_()     # mapped once, before the fork
            q, fq = ctx.Queue(), ctx.Queue()
            for i in range(self.prefetch):
                fq.put(i)
            p = ctx.Process(target=_worker_main, args=(*a, ring, fq, q, self._release), daemon=True)
            p.start()
            self._procs.append(p)
            self._queues.append(q)
            self._free.append(fq)
            self._rings.append(ring)
        return self

    def _raw(self) -> Iterator[Dict]:
        """The workers' batch descriptors in the fixed round-robin order; `frames` are VIEWS into the worker's ring slot, valid until
        `release(b)` hands the slot back."""
        live = list(range(self.workers))
        while live:
            for w in list(live):
                b = self._get(w)
                if b is None:
                    live.remove(w)
                    continue
                b["worker"] = w
                b["frames"] = [self._rings[w][b["slot"]][off:off + 3 * t * h * wd].view(t, h, wd, 3) for off, t, h, wd in b["clips"]]
                yield b

    def _get(self, w: int):
        """Next descriptor of worker w.  A worker that failed sends a _Failure; one that died without a word (killed, segfault)
        is noticed by its exit code - both raise LoaderError here instead of looking like the end of the data."""
        while True:
            try:
                b = self._queues[w].get(timeout=1.0)
            except queue.Empty:
                if self._stop.is_set():
                    return None
                p = self._procs[w]
                if not p.is_alive():
                    try:                            # a last message may still be in the pipe
                        b = self._queues[w].get(timeout=0.5)
                    except queue.Empty:
                        raise LoaderError(f"loader worker {w} died without a message (exit code {p.exitcode})")
                else:
                    continue
            if isinstance(b, _Failure):
                raise LoaderError(f"{b.where} failed:\n{b.text}")
            return b

    def release(self, b: Dict) -> None:
        self._free[b["worker"]].put(b["slot"])

    def raw_batches(self) -> Iterator[Dict]:
        """Host-side batches (tests, CPU consumers): the frames are copied out of the ring and the slot is returned at once."""
        for b in self._raw():
            b["frames"] = [f.clone() for f in b["frames"]]
            self.release(b)
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
            staging = None
            in_flight = []                         # (event, batch descriptor): slots whose upload may still be reading
            try:
                if not self._registered:           # the rings become pinned host memory: uploads go straight from them
                    rt = torch.cuda.cudart()
                    self._registered = True
                    for r in self._rings:          # every ring is tried; the ones that succeeded are remembered for close()
                        if int(rt.cudaHostRegister(r.data_ptr(), r.numel(), 0)) == 0:
                            self._pinned_rings.append(r)
                    self._pinned = len(self._pinned_rings) == len(self._rings)
                for b in self._raw():
                    if self._stop.is_set():
                        break
                    while in_flight and (in_flight[0][0].query() or len(in_flight) >= self.prefetch - 1):
                        ev0, b0 = in_flight.pop(0)
                        ev0.synchronize()
                        self.release(b0)
                    frames = b["frames"]
                    if not self._pinned:           # registration refused: one staging copy through ordinary pinned memory
                        n_bytes = sum(f.numel() for f in frames)
                        if staging is None or staging.numel() < n_bytes:
                            staging = torch.empty(max(n_bytes, self.slot_bytes), dtype=torch.uint8).pin_memory()
                        off, views = 0, []
                        for f in frames:
                            v = staging[off:off + f.numel()].view(f.shape)
                            v.copy_(f)
                            views.append(v)
                            off += (f.numel() + 15) // 16 * 16
                        frames = views
                    with torch.cuda.stream(up):
                        clips = []
                        for v in frames:
                            t, h, w, _c = v.shape
                            d8 = torch.empty(v.shape, dtype=torch.uint8, device=device)
                            d8.copy_(v, non_blocking=True)
                            clip = torch.empty((3, t, h, w), dtype=dtype, device=device)
                            _lib.check(lib.ttv_clip_from_u8(d8.data_ptr(), t, h, w, clip.data_ptr(), code, up.cuda_stream), "ttv_clip_from_u8")
                            clips.append(clip)
                        ev = torch.cuda.Event()
                        ev.record(up)
                    if not self._pinned:
                        ev.synchronize()           # the staging buffer is reused by the next batch
                    in_flight.append((ev, b))
                    item = ({"video": clips, "fps": b["fps"], "__key__": b["__key__"],
                             "token_counts": torch.tensor(b["token_counts"], dtype=torch.int32)}, ev)
                    while not self._stop.is_set():  # a consumer that stopped early must not leave this thread blocked in put()
                        try:
                            out_q.put(item, timeout=0.2)
                            break
                        except queue.Full:
                            pass
                for ev0, b0 in in_flight:
                    ev0.synchronize()
                    self.release(b0)
                in_flight = []
                out_q.put(None)                    # end of data
            except BaseException:
                for ev0, _b0 in in_flight:         # nothing may still be reading a ring when close() unregisters it
                    ev0.synchronize()
                fail = _Failure("loader upload thread", traceback.format_exc())
                while True:                        # the sentinel must arrive: make room if the consumer is not reading
                    try:
                        out_q.put_nowait(fail)
                        break
                    except queue.Full:
                        try:
                            out_q.get_nowait()
                        except queue.Empty:
                            pass

        th = threading.Thread(target=uploader, daemon=True)
        th.start()
        self._uploader, self._upload_q = th, out_q
        while True:
            item = out_q.get()
            if item is None:
                return
            if isinstance(item, _Failure):
                raise LoaderError(f"{item.where} failed:\n{item.text}")
            batch, ev = item
            cur = torch.cuda.current_stream(device)
            cur.wait_event(ev)
            for c in batch["video"]:
                c.record_stream(cur)               # allocated on the upload stream, read on this one
            yield batch

    def close(self) -> None:
        self._stop.set()
        # the upload thread first: it may hold copies in flight out of the rings; drain its queue so a blocked put() returns, join it,
        # and only then unregister the rings
        th, q = getattr(self, "_uploader", None), getattr(self, "_upload_q", None)
        if th is not None:
            while th.is_alive():
                try:
                    while True:
                        q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.2)
            self._uploader = self._upload_q = None
        if getattr(self, "_release", None) is not None:
            self._release.set()
        for p in self._procs:
            p.join(timeout=2)
        for p in self._procs:
            if p.is_alive():
                p.terminate()
        for p in self._procs:
            p.join(timeout=5)
        if self._pinned_rings:                     # every ring that WAS registered, also when a later registration failed
            rt = torch.cuda.cudart()
            if torch.cuda.is_initialized():
                torch.cuda.synchronize()
            for r in self._pinned_rings:
                rt.cudaHostUnregister(r.data_ptr())
        self._pinned_rings = []
        self._procs, self._queues, self._free, self._rings, self._registered = [], [], [], [], False
