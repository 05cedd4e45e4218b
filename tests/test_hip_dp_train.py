"""Data-parallel training step on the HIP path: two ranks (both on cuda:0, gloo for the collective since
one GPU cannot host two RCCL ranks) each run forward/backward on their shard of the clips; after the count-weighted
gradient all-reduce every rank must hold the single-process gradients of the union batch, and one optimiser step must
leave identical weights.  `-m gpu`."""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
SHAPES = [(4, 16, 16), (8, 32, 48), (4, 8, 24)]
COUNTS = [2, 5, 3]


def _model():
    from titok_video_amd.model.titok import TiTok
    from titok_video_amd.synthetic import seeded_titok_state
    cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(
        patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny")))
    m = TiTok(cfg)
    m.load_state_dict(seeded_titok_state(0), strict=True)
    return m.to("cuda:0", torch.float32).train()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from titok_video_amd import dp
        from titok_video_amd.synthetic import synthetic_clips
        from titok_video_amd.train import make_optimizer, training_step
        model = _model()
        all_clips = synthetic_clips(SHAPES, seed=77, dtype=torch.float32, device="cuda:0")
        mine = dp.shard_clips(len(SHAPES), rank, world)          # rank 0: clips 0, 2 ; rank 1: clip 1  (ragged)
        opt = make_optimizer(model)
        loss, gnorm, idx = training_step(model, [all_clips[i] for i in mine], [COUNTS[i] for i in mine], opt)
        torch.cuda.synchronize()
        q.put((rank, {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}, float(gnorm)))   # by value
    finally:
        dist.destroy_process_group()


def test_two_rank_training_step_equals_single_process():
    from titok_video_amd.synthetic import synthetic_clips
    from titok_video_amd.train import make_optimizer, training_step
    # single process, union batch
    model = _model()
    clips = synthetic_clips(SHAPES, seed=77, dtype=torch.float32, device="cuda:0")
    opt = make_optimizer(model)
    loss, gnorm, idx = training_step(model, clips, COUNTS, opt)
    ref = {n: p.detach().cpu() for n, p in model.named_parameters()}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, params, gn in res:
        assert abs(gn - float(gnorm)) < 1e-3 * float(gnorm)
        for n, v in params.items():
            err = float((torch.from_numpy(v) - ref[n]).norm() / (ref[n].norm() + 1e-30))
            assert err < 1e-5, (rank, n, err)
