"""Multi-process (world_size 2, gloo, CPU) checks of the N > 1 path: rank-disjoint clip sharding, codebook histogram
all-reduce == single logger seeing all samples, count-weighted gradient averaging == single-process mean."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from titok_video_amd.codebook import CodebookLogger
from titok_video_amd import dp

G = os.path.join(os.path.dirname(__file__), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = np.load(os.path.join(G, "codebook_kat.npz"))
        sizes = d["sizes"].tolist()
        n = int(d["codebook_size"])
        samples = list(torch.split(torch.from_numpy(d["flat"]), sizes))[-n:]          # what the reference's FIFO holds
        mine = [samples[i] for i in dp.shard_clips(len(samples), rank, world)]
        lg = CodebookLogger(n, world_size=world)       # capacity n/world samples per rank
        lg(mine)
        assert lg.is_score_ready()
        sc = lg.get_scores()                            # all-reduces the int64 histogram over the gloo group
        usage, ent = sc["codebook/usage_percent"], sc["codebook/entropy"]
        # ragged gradient averaging: rank r holds (r+1) clips; local grad = mean over its clips of per-clip grads
        g = torch.Generator().manual_seed(5)
        per_clip = torch.randn(3, 7, generator=g)                # 3 clips in the union batch
        own = per_clip[:1] if rank == 0 else per_clip[1:]
        grad = own.mean(0).clone()
        total = dp.allreduce_mean_by_count([grad], own.shape[0])
        gathered = dp.gather_variable(torch.arange(rank + 2, dtype=torch.int32))
        q.put((rank, usage, ent, grad.numpy(), total, [t.tolist() for t in gathered], per_clip.mean(0).numpy()))
    finally:
        dist.destroy_process_group()


def test_world2_histogram_and_gradient_semantics():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = np.load(os.path.join(G, "codebook_kat.npz"))
    for rank, usage, ent, grad, total, gathered, ref_grad in res:
        assert abs(usage - float(d["usage"])) < 1e-4          # == the reference's single-process logger
        assert abs(ent - float(d["entropy"])) < 1e-5
        np.testing.assert_allclose(grad, ref_grad, rtol=1e-6, atol=1e-6)   # == mean over the union batch
        assert total == 3
        assert gathered == [[0, 1], [0, 1, 2]]


def test_single_process_logger_matches_reference_fixture():
    d = np.load(os.path.join(G, "codebook_kat.npz"))
    sizes = d["sizes"].tolist()
    n = int(d["codebook_size"])
    lg = CodebookLogger(n)
    assert lg.get_scores() is None
    lg(list(torch.split(torch.from_numpy(d["flat"]), sizes)))    # 80 samples through a FIFO of 64
    assert lg.is_score_ready()
    sc = lg.get_scores()
    assert abs(sc["codebook/usage_percent"] - float(d["usage"])) < 1e-4
    assert abs(sc["codebook/entropy"] - float(d["entropy"])) < 1e-5
    assert lg.codebook_indices == [] and not lg.is_score_ready()


def test_shard_clips_is_a_partition():
    for n, w in [(32, 8), (5, 2), (3, 4), (0, 2)]:
        parts = [dp.shard_clips(n, r, w) for r in range(w)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
