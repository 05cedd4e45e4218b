"""titok_video_amd/loader.py on the CPU: the worker processes' batch stream (uint8 frames in shared memory, token-budget batching with
the reference's policy, video_dataset.py:130-172) is deterministic and equals what one process computes from the same shards."""
import os

import numpy as np
import torch

from titok_video_amd.data import dynamic_batches
from titok_video_amd.loader import ShardBatchLoader, raw_shard_samples
from titok_video_amd.shards import shard_samples, write_synthetic_shards


def _take(it, n):
    out = []
    for b in it:
        out.append(b)
        if len(out) == n:
            break
    return out


def test_worker_batches_are_deterministic_and_match_one_process(tmp_path):
    paths = write_synthetic_shards(str(tmp_path), 4, 16, min_grid=(4, 16, 16), max_grid=(8, 32, 32), seed=3)
    kw = dict(patch=(4, 8, 8), token_range=(1, 16), seq_len=96, seed=7, epochs=1, drop_last=True)
    runs = []
    for _ in range(2):
        ld = ShardBatchLoader(paths, rank=1, world_size=2, workers=2, **kw).start()
        runs.append(list(ld.raw_batches()))
        ld.close()
    assert len(runs[0]) == len(runs[1]) > 2
    for a, b in zip(*runs):
        assert a["__key__"] == b["__key__"] and a["token_counts"] == b["token_counts"]
        assert all(torch.equal(x, y) for x, y in zip(a["frames"], b["frames"]))
    # rank 1 of 2 owns shards 1 and 3; worker 0 reads shard 1, worker 1 shard 3 (seeds 7 and 7 + 1009); batches alternate
    mine = [p for i, p in enumerate(sorted(paths)) if i % 2 == 1]
    per_worker = [list(dynamic_batches(raw_shard_samples([mine[w]], 1), (4, 8, 8), (1, 16), 96, seed=7 + 1009 * w, drop_last=True)) for w in range(2)]
    want = []
    for i in range(max(len(p) for p in per_worker)):
        for w in range(2):
            if i < len(per_worker[w]):
                want.append(per_worker[w][i])
    assert [b["__key__"] for b in runs[0]] == [b["__key__"] for b in want]
    for got, ref in zip(runs[0], want):
        assert got["token_counts"] == ref["token_counts"].tolist()
        for f, v in zip(got["frames"], ref["video"]):
            assert f.dtype == torch.uint8 and torch.equal(f.permute(3, 0, 1, 2), v)
            rows = sum(np.prod([d // p for d, p in zip(v.shape[1:], (4, 8, 8))]) for v in ref["video"]) + sum(got["token_counts"])
            assert rows <= 96
    # and the pixel values are those the host-side reader normalises (shards.shard_samples): u8 / 127.5 - 1
    first = next(iter(shard_samples([mine[0]], dtype=torch.float32)))
    f0 = per_worker[0][0]["video"][0]
    assert first["__key__"] == per_worker[0][0]["__key__"][0]
    assert torch.equal(first["video"], f0.to(torch.float32) / 127.5 - 1.0)


def test_host_thread_limit_respects_the_container_share():
    """train.limit_host_threads: never more threads than the affinity mask / cgroup quota grants, never more than asked for."""
    import os

    import torch

    from titok_video_amd.train import host_cpu_share, limit_host_threads
    before = torch.get_num_threads()
    try:
        share = host_cpu_share()
        assert 1 <= share <= (os.cpu_count() or 1)
        n = limit_host_threads(3)
        assert n == torch.get_num_threads() and 1 <= n <= min(3, share)
    finally:
        torch.set_num_threads(before)


def test_a_failing_worker_raises_in_the_consumer_instead_of_ending_the_data(tmp_path):
    """a corrupt shard (or any exception in a worker) used to look like a normal end of the epoch.  Shard 1 is
    truncated in the middle of a member; the consumer must get LoaderError carrying the worker's traceback, after the batches of
    the healthy worker that were already dealt."""
    import pytest

    from titok_video_amd.loader import LoaderError
    paths = write_synthetic_shards(str(tmp_path), 2, 12, min_grid=(4, 16, 16), max_grid=(8, 32, 32), seed=5)
    size = os.path.getsize(paths[1])
    with open(paths[1], "r+b") as f:
        f.truncate(size // 2 + 77)
    ld = ShardBatchLoader(paths, workers=2, patch=(4, 8, 8), token_range=(1, 16), seq_len=96, seed=1, epochs=1, drop_last=True).start()
    try:
        with pytest.raises(LoaderError) as e:
            list(ld.raw_batches())
        assert "loader worker" in str(e.value) and "Traceback" in str(e.value)
    finally:
        ld.close()


def test_a_killed_worker_is_noticed(tmp_path):
    """A worker that dies without a message (SIGKILL: the OOM killer) raises too - by its exit code."""
    import signal

    import pytest

    from titok_video_amd.loader import LoaderError
    paths = write_synthetic_shards(str(tmp_path), 1, 8, min_grid=(4, 16, 16), max_grid=(8, 32, 32), seed=6)
    ld = ShardBatchLoader(paths, workers=1, patch=(4, 8, 8), token_range=(1, 16), seq_len=96, seed=1, epochs=None, drop_last=True, prefetch=2).start()
    try:
        it = ld._raw()
        next(it)                                   # the worker is alive and blocked on a free slot (prefetch 2, nothing released)
        os.kill(ld._procs[0].pid, signal.SIGKILL)
        with pytest.raises(LoaderError) as e:
            for _ in range(8):
                next(it)
        assert "died without a message" in str(e.value)
    finally:
        ld.close()
