"""BASELINE config #3 as a test: configs/tiny.yaml training on synthetic WebDataset-style shards, data parallel.

Four small shards are written; two ranks (both on cuda:0, gloo for the collective: one GPU cannot host two RCCL ranks) each feed the
reference's generator step (train.py:65-83: forward, L1, backward, clip 1.0, AdamW) from their own shards through the process-based
loader (titok_video_amd/loader.py: worker processes -> shared memory -> pinned staging -> GPU normalisation), gradients reduced
behind the backward (dp.GradReducer), token indices logged (CodebookLogger).  One process then runs the same steps on the UNION of
the two ranks' batches.  The first step's reduced gradients agree element by element (2e-5 of the tensor's largest); after N steps the parameters agree to 1e-4 (relative; an element is exempt only where its Adam second moment says its gradient is summation noise - see the test) and the
codebook histogram exactly.  (The discriminator step of tools/train_dp.py draws fresh noise per rank, so the comparable quantity is
the generator step.)  `-m gpu`."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
STEPS = 3
LOADER = dict(patch=(4, 8, 8), token_range=(1, 16), seq_len=160, workers=2, epochs=None, drop_last=True)


def _model():
    from titok_video_amd.model.titok import TiTok
    from titok_video_amd.synthetic import seeded_titok_state
    cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(
        patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny")))
    m = TiTok(cfg)
    m.load_state_dict(seeded_titok_state(0), strict=True)
    return m.to("cuda:0", torch.float32).train()


def _grads(model):
    return {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters() if p.grad is not None}


def _adam_v(model, opt):
    return {n: opt.state[p]["exp_avg_sq"].detach().cpu().numpy().copy() for n, p in model.named_parameters() if p in opt.state}


def _rank_worker(rank, world, port, paths, q):
    from titok_video_amd.loader import ShardBatchLoader
    loader = ShardBatchLoader(paths, rank, world, seed=100 + rank, **LOADER).start()       # before the first GPU call
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from titok_video_amd.codebook import CodebookLogger
        from titok_video_amd.data import equal_steps
        from titok_video_amd.train import make_optimizer, training_step
        model = _model()
        opt = make_optimizer(model)
        logger = CodebookLogger(4375, world_size=world)
        it = iter(equal_steps(loader.batches("cuda:0", torch.float32)))
        keys, g0 = [], None
        for st in range(STEPS):
            b = next(it)
            counts = b["token_counts"].tolist()
            loss, gnorm, idx = training_step(model, b["video"], counts, opt)
            if st == 0:
                g0 = _grads(model)         # reduced + clipped gradients of the step both runs take from identical weights
            logger(torch.split(idx, counts))
            keys.append(list(b["__key__"]))
        torch.cuda.synchronize()
        q.put((rank, {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}, logger.histogram().cpu().numpy(), keys,
               g0, _grads(model), _adam_v(model, opt)))
    finally:
        loader.close()
        dist.destroy_process_group()


def _single_worker(paths, q):
    """The union run in a fresh process (the loader workers must be forked before the process touches the GPU)."""
    from titok_video_amd.loader import ShardBatchLoader
    loaders = [ShardBatchLoader(paths, r, 2, seed=100 + r, **LOADER).start() for r in range(2)]
    from titok_video_amd.codebook import CodebookLogger
    from titok_video_amd.train import make_optimizer, training_step
    model = _model()
    opt = make_optimizer(model)
    logger = CodebookLogger(4375)
    its = [iter(ld.batches("cuda:0", torch.float32)) for ld in loaders]
    keys, g0 = [], None
    for st in range(STEPS):
        bs = [next(it) for it in its]
        clips = [c for b in bs for c in b["video"]]
        counts = [k for b in bs for k in b["token_counts"].tolist()]
        loss, gnorm, idx = training_step(model, clips, counts, opt)
        if st == 0:
            g0 = _grads(model)
        logger(torch.split(idx, counts))
        keys.append([list(b["__key__"]) for b in bs])
    torch.cuda.synchronize()
    q.put(("single", {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}, logger.histogram().cpu().numpy(), keys,
           g0, _grads(model), _adam_v(model, opt)))
    for ld in loaders:
        ld.close()


def test_two_rank_shard_training_equals_one_process_on_the_union(tmp_path):
    from titok_video_amd.shards import write_synthetic_shards
    paths = write_synthetic_shards(str(tmp_path), 4, 24, min_grid=(4, 16, 16), max_grid=(8, 32, 48), seed=5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, paths, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, params, hist, keys, g0, g_last, v = q.get(timeout=600)
        res[rank] = (params, hist, keys, g0, g_last, v)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    one = ctx.Process(target=_single_worker, args=(paths, q))
    one.start()
    tag, params1, hist1, keys1, g0_1, glast_1, v_1 = q.get(timeout=600)
    one.join(timeout=120)
    assert one.exitcode == 0 and tag == "single"
    # the same clips went through both runs, step by step
    for st in range(STEPS):
        assert keys1[st] == [res[0][2][st], res[1][2][st]]
    assert set(res[0][2][0]).isdisjoint(res[1][2][0])                      # rank-disjoint shards
    # (1) DP == single process, asserted where it can be asserted tightly: the first step starts from identical weights, so the reduced,
    # clipped gradients of the two-rank run (sum(count * grad) / sum(count) over the ranks, dp.GradReducer) must be the gradients of the
    # one-process step on the union - every element, no exemption.  Tolerance 2e-5 of the tensor's largest gradient: the fp32 kernels
    # accumulate in different orders (atomics, and two half-batches instead of one batch), ~1e-7 relative per sum.
    for n in g0_1:
        assert np.array_equal(res[0][3][n], res[1][3][n]), n
        gs = float(np.abs(g0_1[n]).max()) + 1e-30
        gd = np.abs(res[0][3][n] - g0_1[n])
        assert float(gd.max()) <= 2e-5 * gs, ("step-0 gradient", n, float(gd.max()), gs)
    # (2) the weights after STEPS AdamW steps: both ranks identical, and equal to the single-process weights within 1e-4 of the tensor's
    # scale - EXCEPT where AdamW's normalised update is noise: an element whose gradient is itself of the size of the summation noise
    # (|g| ~ 1e-7 of the tensor's gradients, sqrt(v) tiny) moves by up to lr (1 + wd |p|) per step in a direction the noise picks.  Such an
    # element is exempt only when the single-process second moment says so (sqrt(v) below 1e-4 of the tensor's largest); any other
    # disagreement fails and prints both runs' gradient and v of the element.
    lr = 1e-4                                                              # train.make_optimizer's default (configs/tiny.yaml)
    for n in params1:
        assert np.array_equal(res[0][0][n], res[1][0][n]), n
        scale = float(np.abs(params1[n]).max()) + 1e-12
        diff = np.abs(res[0][0][n] - params1[n])
        bad = np.argwhere(diff >= 1e-4 * scale)
        if len(bad):
            rv = np.sqrt(v_1[n])
            rv_max = float(rv.max()) + 1e-30
            for ix in map(tuple, bad):
                report = (f"{n}{list(ix)}: |dp - single| {diff[ix]:.3e} (scale {scale:.3e}); last gradient dp {res[0][4][n][ix]:.3e} single "
                          f"{glast_1[n][ix]:.3e}; sqrt(v) dp {np.sqrt(res[0][5][n][ix]):.3e} single {rv[ix]:.3e} (tensor max {rv_max:.3e})")
                print("noise-dominated AdamW element: " + report)
                assert rv[ix] <= 1e-4 * rv_max, "parameters differ where the gradient is NOT noise: " + report
                assert diff[ix] <= 2.2 * lr * STEPS, report
    # codebook usage: the sum of the ranks' histograms is the single logger's histogram, exactly
    assert np.array_equal(res[0][1] + res[1][1], hist1) and int(hist1.sum()) > 0
