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


def _overlap_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from titok_video_amd import dp
        from titok_video_amd.synthetic import synthetic_clips
        from titok_video_amd.train import make_optimizer, training_step
        all_clips = synthetic_clips(SHAPES, seed=77, dtype=torch.float32, device="cuda:0")
        mine = dp.shard_clips(len(SHAPES), rank, world)
        out = {}
        for overlap in (False, True):
            model = _model()
            opt = make_optimizer(model)
            # gradients only: clip / step afterwards would hide nothing but make the comparison about the optimizer
            opt.zero_grad(set_to_none=True)
            from titok_video_amd.train import _reducer_for, _towers, l1_reconstruction_loss
            red = _reducer_for(model, None, overlap)
            if red is not None:
                red.attach(*_towers(model))
                red.begin_step(len(mine))
            clips = [all_clips[i] for i in mine]
            recon, info = model(clips, [COUNTS[i] for i in mine])
            l1_reconstruction_loss(recon, clips).backward()
            params = [p for p in model.parameters() if p.grad is not None]
            if red is not None:
                red.detach(*_towers(model))
                red.finish()
                assert red.slices == 2 * (4 + 1)            # per tower: one slice per layer + the head / tail block
            else:
                dp.allreduce_mean_by_count([p.grad for p in params], len(mine))
            torch.cuda.synchronize()
            out[overlap] = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}
        # the reduction arithmetic alone, on the SAME local buffer: slices through the reducer vs one bucket through the sequential path
        g = torch.Generator().manual_seed(50 + rank)
        local = torch.randn(100_003, generator=g).to("cuda:0")
        a, b = local.clone(), local.clone()
        red = dp.GradReducer(torch.device("cuda:0"))
        red.begin_step(rank + 1)
        ev = torch.cuda.Event()
        ev.record()
        for lo, hi in ((60_000, 100_003), (0, 60_000)):
            red.reduce_slice(a, lo, hi, ev)
        red.finish()
        dp.allreduce_mean_by_count([b], rank + 1)
        torch.cuda.synchronize()
        out["arith_equal"] = bool(torch.equal(a, b))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_overlapped_gradient_reduction_equals_the_sequential_one_bit_for_bit():
    """dp.GradReducer (layer slices reduced in place from the flat buffer, behind the backward) against dp.allreduce_mean_by_count
    (after the backward, concatenated buckets): the same multiply - sum over ranks - divide per element, so bit-identical fp32
    weight gradients on both ranks (vector parameters carry the backward's own atomic-order noise), ragged clip counts (2 + 1)."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert out.pop("arith_equal")                        # same buffer in, bit-identical out: multiply - sum over ranks - divide
        assert set(out[False]) == set(out[True]) and len(out[True]) == 76
        for n in out[False]:
            # two separate fp32 backward runs: the float32 gradient kernels accumulate with atomics (run-to-run last-bit noise), so
            # the end-to-end comparison carries that noise; the reduction itself is checked bit for bit above
            scale = float(np.abs(out[False][n]).max()) + 1e-12
            assert float(np.abs(out[False][n] - out[True][n]).max()) < 1e-4 * scale, (rank, n)
    # and both ranks hold the same reduced gradients
    for n in res[0][1][True]:
        assert np.array_equal(res[0][1][True][n], res[1][1][True][n]), n       # an all-reduce leaves the same bits on every rank


def _rccl_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # "nccl" is RCCL on ROCm
    try:
        from titok_video_amd import dp
        from titok_video_amd.model.titok import TiTok
        from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
        from titok_video_amd.train import make_optimizer, training_step
        cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(
            patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny")))
        out = {}
        for overlap in (True, False):
            model = TiTok(cfg)
            model.load_state_dict(seeded_titok_state(0), strict=True)
            model = model.to(dev, torch.float32).train()
            all_clips = synthetic_clips(SHAPES, seed=77, dtype=torch.float32, device=dev)
            mine = dp.shard_clips(len(SHAPES), rank, world)
            opt = make_optimizer(model)
            loss, gnorm, idx = training_step(model, [all_clips[i] for i in mine], [COUNTS[i] for i in mine], opt, overlap=overlap)
            torch.cuda.synchronize(dev)
            out[overlap] = ({n: p.detach().cpu().numpy() for n, p in model.named_parameters()}, float(gnorm))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: this box has fewer than two")
def test_two_rank_training_step_over_rccl_equals_single_process():
    """The same union-batch equality as above with one GPU per rank and backend "nccl" (RCCL over xGMI), overlapped and sequential
    reduction.  Skipped on one-GPU boxes (every box this code was developed on): it exists for the first multi-GPU lease."""
    from titok_video_amd.synthetic import synthetic_clips
    from titok_video_amd.train import make_optimizer, training_step
    model = _model()
    clips = synthetic_clips(SHAPES, seed=77, dtype=torch.float32, device="cuda:0")
    opt = make_optimizer(model)
    loss, gnorm, idx = training_step(model, clips, COUNTS, opt)
    ref = {n: p.detach().cpu() for n, p in model.named_parameters()}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        for overlap, (params, gn) in out.items():
            assert abs(gn - float(gnorm)) < 1e-3 * float(gnorm), (rank, overlap)
            for n, v in params.items():
                err = float((torch.from_numpy(v) - ref[n]).norm() / (ref[n].norm() + 1e-30))
                assert err < 1e-5, (rank, overlap, n, err)


def test_communication_stream_path_on_one_gpu_defers_gradients():
    """the non-gloo branch of dp.GradReducer (communication stream, per-layer events, gradients produced on that
    stream) never ran on the one-GPU boxes.  `simulate=True` runs exactly that branch without a process group (the collective of a
    world of one is the identity).  Checked here: (1) the tower gradients are NOT handed to autograd (p.grad stays None until
    finish()), (2) finish() delivers them behind the stream join, equal to the plain backward's up to the fp32 kernels' atomic-order
    noise, (3) a gradient that is already there is accumulated into, not overwritten or raced with (micro-batch accumulation)."""
    from titok_video_amd import dp
    from titok_video_amd.synthetic import synthetic_clips
    from titok_video_amd.train import _towers, l1_reconstruction_loss
    clips = synthetic_clips(SHAPES + [(4, 16, 16)], seed=77, dtype=torch.float32, device="cuda:0")
    counts = COUNTS + [4]                                   # 4 clips: count * g / count is exact in fp32

    def backward(model, red):
        towers = _towers(model)
        if red is not None:
            red.attach(*towers)
            red.begin_step(len(clips))
        try:
            recon, _ = model(clips, counts)
            l1_reconstruction_loss(recon, clips).backward()
        finally:
            if red is not None:
                red.detach(*towers)

    plain = _model()
    backward(plain, None)
    torch.cuda.synchronize()
    ref = {n: p.grad.clone() for n, p in plain.named_parameters()}

    model = _model()
    red = dp.GradReducer("cuda:0", simulate=True)
    assert red.on_comm_stream
    backward(model, red)
    assert all(p.grad is None for p in model.parameters())             # (1) nothing reached autograd's accumulation
    assert red.slices == 2 * (4 + 1) and len(red._deferred) == 76      # 4 layer slices + head/tail per tower; every parameter deferred
    red.finish()
    torch.cuda.synchronize()
    for n, p in model.named_parameters():                             # (2)
        scale = float(ref[n].abs().max()) + 1e-12
        assert float((p.grad - ref[n]).abs().max()) < 1e-4 * scale, n
    first = {n: p.grad.clone() for n, p in model.named_parameters()}
    backward(model, red)                                               # (3) second micro-batch on top of existing gradients
    red.finish()
    torch.cuda.synchronize()
    for n, p in model.named_parameters():
        scale = float(ref[n].abs().max()) + 1e-12
        assert float((p.grad - (first[n] + ref[n])).abs().max()) < 2e-4 * scale, n
