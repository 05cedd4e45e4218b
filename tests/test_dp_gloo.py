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


def _ragged_worker(rank, world, port, q):
    """Ranks that receive different numbers of clips per step (token-budget batching): the logger's readiness decision and the
    end of the epoch must be collective, or one rank enters an all-reduce the other skips (a hang on RCCL)."""
    from titok_video_amd.data import SyntheticClipStream, dynamic_batches, equal_steps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 8
        lg = CodebookLogger(n, world_size=world)               # capacity 4 samples per rank
        g = torch.Generator().manual_seed(100 + rank)
        fired = []
        for step in range(6):
            k = 1 if rank == 0 else 2                          # rank 1 fills its FIFO twice as fast
            lg([torch.randint(0, n, (5,), generator=g, dtype=torch.int32) for _ in range(k)])
            sc = lg.get_scores()                               # every rank makes the same collective calls every step
            fired.append(sc is not None)
        # epoch driver: a bounded stream sharded i % world gives the ranks different numbers of batches
        stream = SyntheticClipStream(min_grid=(4, 16, 16), max_grid=(8, 32, 32), dtype=torch.float32, seed=3, rank=rank, world_size=world,
                                     length=13)
        mine = list(dynamic_batches(stream, (4, 8, 8), (1, 8), 96, seed=rank, drop_last=True))
        steps = 0
        for _b in equal_steps(iter(mine)):
            t = torch.ones(1)
            dist.all_reduce(t)                                 # stands for the per-step gradient all-reduce
            steps += 1
        # the same through the CPU-side control group a RCCL run would create for the flag (forced here: the main group is gloo already)
        steps_ctl = sum(1 for _b in equal_steps(iter(mine), _force_control_group=True))
        assert steps_ctl == steps
        q.put((rank, fired, len(mine), steps))
    finally:
        dist.destroy_process_group()


def test_world2_ragged_steps_keep_collectives_matched():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, fired0, n0, s0), (_, fired1, n1, s1) = res
    assert fired0 == fired1                                  # both ranks report on the same steps ...
    assert fired0.index(True) == 3                           # ... the first time when the SLOWER rank (1 clip per step) has 4 samples
    assert s0 == s1 == min(n0, n1)                           # the epoch ends together, at the shorter rank's length


def test_dynamic_batches_drop_last_matches_the_reference_policy():
    from titok_video_amd.data import SyntheticClipStream, dynamic_batches
    stream = SyntheticClipStream(min_grid=(4, 16, 16), max_grid=(8, 32, 32), dtype=torch.float32, seed=1, length=9)
    keep = list(dynamic_batches(stream, (4, 8, 8), (1, 8), 96, seed=0))
    drop = list(dynamic_batches(stream, (4, 8, 8), (1, 8), 96, seed=0, drop_last=True))
    assert len(drop) == len(keep) - 1                        # the clips left over at the end of the stream are never emitted ...
    for a, b in zip(drop, keep):                             # ... everything before them is unchanged
        assert a["__key__"] == b["__key__"] and torch.equal(a["token_counts"], b["token_counts"])
