"""Synthetic shard stream + token-budget dynamic batching (host logic, CPU)."""
import itertools
import math

import torch

from titok_video_amd.data import SyntheticClipStream, dynamic_batches

PATCH = (4, 8, 8)


def test_dynamic_batches_respect_the_token_budget_and_contract():
    stream = SyntheticClipStream(dtype=torch.float32, seed=3, length=40)
    batches = list(dynamic_batches(stream, PATCH, (1, 128), 6144, seed=1, max_grid=(16, 168, 168)))
    assert sum(len(b["video"]) for b in batches) == 40
    for b in batches:
        assert set(b) == {"video", "fps", "__key__", "token_counts"}
        assert b["token_counts"].dtype == torch.int32 and len(b["token_counts"]) == len(b["video"])
        rows = sum(math.prod(x // p for x, p in zip(v.shape[1:], PATCH)) + int(k) for v, k in zip(b["video"], b["token_counts"]))
        assert rows <= 6144
        for v, k in zip(b["video"], b["token_counts"]):
            assert v.shape[0] == 3 and all(s % p == 0 for s, p in zip(v.shape[1:], PATCH))
            assert 8 <= v.shape[1] <= 16 and 128 <= v.shape[2] <= 168 and 128 <= v.shape[3] <= 168
            assert max(v.shape[2], v.shape[3]) <= 2 * min(v.shape[2], v.shape[3])
            assert 1 <= int(k) <= 128 and float(v.min()) >= -1 and float(v.max()) <= 1
    # greedy policy: a batch is closed only when the next sample would not fit
    assert all(len(b["video"]) >= 1 for b in batches)


def test_stream_is_deterministic_and_rank_disjoint():
    a = [s["__key__"] for s in itertools.islice(SyntheticClipStream(seed=5, dtype=torch.float32), 6)]
    b = [s["__key__"] for s in itertools.islice(SyntheticClipStream(seed=5, dtype=torch.float32), 6)]
    assert a == b
    r0 = list(SyntheticClipStream(seed=5, rank=0, world_size=2, length=10, dtype=torch.float32))
    r1 = list(SyntheticClipStream(seed=5, rank=1, world_size=2, length=10, dtype=torch.float32))
    keys = sorted([s["__key__"] for s in r0] + [s["__key__"] for s in r1])
    assert keys == [f"synthetic_{i:08d}" for i in range(10)]
    full = list(SyntheticClipStream(seed=5, length=10, dtype=torch.float32))
    assert torch.equal(full[3]["video"], r1[1]["video"])        # sample 3 regenerated identically by rank 1
