"""Synthetic tar shards (titok_video_amd/shards.py): write / read round trip, rank-disjoint sharding, the batch dict the hot path
consumes (reference dataset/video_dataset.py:160-164) through dynamic batching with the reference's drop-last policy."""
import tarfile

import torch

from titok_video_amd.data import dynamic_batches
from titok_video_amd.shards import shard_samples, write_synthetic_shards


def test_shards_round_trip_and_rank_split(tmp_path):
    paths = write_synthetic_shards(str(tmp_path), n_shards=4, clips_per_shard=3, min_grid=(4, 16, 16), max_grid=(8, 32, 32), seed=5)
    assert len(paths) == 4
    with tarfile.open(paths[0]) as tar:
        names = tar.getnames()
    assert len(names) == 6 and names[0].endswith(".npy") and names[1].endswith(".json")      # {__key__}.npy + {__key__}.json per sample
    all_keys = [s["__key__"] for s in shard_samples(paths, dtype=torch.float32)]
    assert len(all_keys) == 12 and len(set(all_keys)) == 12
    r0 = [s["__key__"] for s in shard_samples(paths, rank=0, world_size=2, dtype=torch.float32)]
    r1 = [s["__key__"] for s in shard_samples(paths, rank=1, world_size=2, dtype=torch.float32)]
    assert not set(r0) & set(r1) and sorted(r0 + r1) == sorted(all_keys)                     # whole shards per rank, disjoint, complete
    s = next(iter(shard_samples(paths, dtype=torch.float32)))
    v = s["video"]
    assert v.shape[0] == 3 and v.dtype == torch.float32 and float(v.min()) >= -1.0 and float(v.max()) <= 1.0
    assert all(x % p == 0 for x, p in zip(v.shape[1:], (4, 8, 8)))
    # deterministic: the same seed gives the same bytes
    again = write_synthetic_shards(str(tmp_path / "b"), n_shards=1, clips_per_shard=3, min_grid=(4, 16, 16), max_grid=(8, 32, 32), seed=5)
    a = next(iter(shard_samples(again, dtype=torch.float32)))
    assert torch.equal(a["video"], v)


def test_shards_feed_dynamic_batches(tmp_path):
    paths = write_synthetic_shards(str(tmp_path), n_shards=2, clips_per_shard=5, min_grid=(4, 16, 16), max_grid=(8, 32, 32), seed=1)
    batches = list(dynamic_batches(shard_samples(paths, dtype=torch.bfloat16), (4, 8, 8), (1, 8), 96, seed=0, drop_last=True))
    assert batches
    for b in batches:
        assert set(b) == {"video", "fps", "__key__", "token_counts"} and b["token_counts"].dtype == torch.int32
        rows = sum(v.shape[1] // 4 * (v.shape[2] // 8) * (v.shape[3] // 8) + int(k) for v, k in zip(b["video"], b["token_counts"]))
        assert rows <= 96
