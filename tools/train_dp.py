#!/usr/bin/env python3
"""BASELINE config #3: configs/tiny.yaml training on synthetic WebDataset-style shards, data-parallel over the GPUs of one node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/train_dp.py --steps 20
    python tools/train_dp.py --steps 20                       (single process)

One process per GPU, backend "nccl" (= RCCL over xGMI).  Every rank writes nothing: rank 0 writes the seeded shards to a scratch
directory first (or --shards points at existing ones), then every rank reads the shards it owns (shard i -> rank i % world), batches
them under the reference's token budget (train_seq_len 6144, token_range [1, 128]: configs/tiny.yaml:57,65) and runs `--steps` steps
of the reference's generator + discriminator step (train.py:64-107; L1 + relativistic GAN term, LPIPS off: no network) with the
gradient all-reduce overlapped with the backward (titok_video_amd.dp.GradReducer) and the codebook-usage histogram all-reduced at the
end.  The epoch ends collectively when the first rank runs out of batches.  Rank 0 prints ONE JSON line in bench.py's shape.
`python tools/train_dp.py --gpus N` without a launcher starts its own N ranks (as bench.py does) before touching the GPU."""
import argparse
import json
import os
import sys
import tempfile
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fan_out_envs(n, port):
    """Environments of the n ranks this script starts by itself: torchrun's variables plus ONE run token shared by all of them (it names
    the shard directory rank 0 writes and the others wait for; a token per rank - the clock read once per Popen - left every rank but 0
    waiting for a directory that never came, round 4)."""
    token = "%d_%d_%d" % (os.getpid(), port, int(time.time() * 1e3))
    return [dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 HSA_ENABLE_IPC_MODE_LEGACY="0", TTV_RUN_TOKEN=token) for r in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="without a launcher (WORLD_SIZE unset): start this many ranks, one per GPU")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shards", default="")
    ap.add_argument("--n-shards", type=int, default=0, help="default 2 per rank")
    ap.add_argument("--clips-per-shard", type=int, default=64)
    ap.add_argument("--seq-len", type=int, default=6144)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--workers", type=int, default=2, help="loader worker processes per rank (the reference uses 3, video_dataset.py:211)")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--repeat-first-batch", action="store_true", help="diagnostics: every step trains on the first batch (plans cached, loader idle)")
    ap.add_argument("--preload", action="store_true", help="diagnostics: fetch every batch of the run before the first step (loader idle during the steps)")
    ap.add_argument("--torch-profile", default="", help="diagnostics: torch.profiler tables (CPU + GPU) of the timed steps of rank 0 -> this file")
    ap.add_argument("--phase-times", action="store_true", help="diagnostics: host milliseconds per step spent waiting for the loader / building plans / in the step")
    ap.add_argument("--host-profile", default="", help="write a cProfile summary of the timed steps of rank 0 to this file (diagnostics)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:      # own fan-out, before anything touches the GPU (reference: L.Trainer(devices=N), train.py:270-280)
        import socket
        import subprocess
        if args.backend != "gloo" and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"train_dp.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible (--backend gloo rehearses all ranks on cuda:0)")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env) for env in fan_out_envs(args.gpus, port)]
        codes = [p.wait() for p in procs]
        if any(codes):
            raise SystemExit(f"train_dp.py --gpus {args.gpus}: rank exit codes {codes}")
        return

    if os.environ.get("TTV_HANG_DUMP"):         # diagnostics: every thread's stack to stderr after that many seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["TTV_HANG_DUMP"]), exit=True)
    import torch.distributed as dist
    from titok_video_amd import dp
    from titok_video_amd.codebook import CodebookLogger
    from titok_video_amd.data import dynamic_batches, equal_steps
    from titok_video_amd.model.losses import ReconstructionLoss
    from titok_video_amd.model.titok import TiTok
    from titok_video_amd.loader import ShardBatchLoader
    from titok_video_amd.shards import write_synthetic_shards
    from titok_video_amd.synthetic import seeded_titok_state, seeded_tower_state
    from titok_video_amd.train import freeze_python_gc, gan_training_step, limit_host_threads, make_discriminator_optimizer, make_optimizer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = args.backend == "gloo"                    # CPU-side collective, every rank on cuda:0 (one-GPU boxes)
    dev_index = 0 if rehearsal else local_rank
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    # ---- shards (rank 0 writes, everybody reads its own) and the loader's worker processes: BEFORE this process touches the GPU
    # (the workers are forked; the ranks meet on a marker file, not on a collective)
    # The directory is unique per run: the launcher of the ranks (this script's own fan-out, or whatever started them) hands every
    # rank the same TTV_RUN_TOKEN; under torchrun, where MASTER_PORT is the constant 29500 and may be reused by the next run, the
    # token falls back to TORCHELASTIC_RUN_ID + the launcher's pid.  Rank 0 writes the shards into a temporary sibling directory and
    # renames it into place (atomic), then writes the token into the marker; the other ranks wait for a marker holding THEIR token, so
    # a directory left behind by an earlier run is never read while rank 0 rewrites it.  Rank 0 removes it at exit.
    token = os.environ.get("TTV_RUN_TOKEN") or "%s_%s_%d" % (os.environ.get("TORCHELASTIC_RUN_ID", "run"), os.environ.get("MASTER_PORT", "single"),
                                                            os.getppid() if world > 1 else os.getpid())
    shard_dir = args.shards or os.path.join(tempfile.gettempdir(), "ttv_shards_%s" % token)
    n_shards = args.n_shards or 2 * world
    marker = os.path.join(shard_dir, ".written")
    if not args.shards:
        if rank == 0:
            import atexit
            import shutil
            shutil.rmtree(shard_dir, ignore_errors=True)            # a stale directory of the same name (token reuse): marker goes first
            staging = shard_dir + ".writing.%d" % os.getpid()
            shutil.rmtree(staging, ignore_errors=True)
            write_synthetic_shards(staging, n_shards, args.clips_per_shard, seed=11)
            with open(os.path.join(staging, ".written"), "w") as f:
                f.write(token)
            os.rename(staging, shard_dir)
            atexit.register(shutil.rmtree, shard_dir, ignore_errors=True)
        else:
            t_wait = time.time()
            while True:
                try:
                    if open(marker).read() == token:
                        break
                except OSError:
                    pass
                if time.time() - t_wait > 120:
                    raise SystemExit("train_dp.py: rank 0 never finished writing the shards (waited 120 s for %s to hold this run's token)" % marker)
                time.sleep(0.05)
    paths = sorted(os.path.join(shard_dir, f) for f in os.listdir(shard_dir) if f.endswith(".tar"))[:n_shards]
    loader = ShardBatchLoader(paths, rank, world, patch=(4, 8, 8), token_range=(1, 128), seq_len=args.seq_len, seed=100 + rank,
                              workers=args.workers, epochs=None, drop_last=True).start()

    if not rehearsal and torch.cuda.device_count() <= dev_index:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        from titok_video_amd.data import setup_control_group
        setup_control_group()                     # collective over the world: the gloo group of equal_steps' control flag

    # ---- model, loss module (discriminator), optimisers: configs/tiny.yaml
    levels = [7, 5, 5, 5, 5]
    cfg = SimpleNamespace(
        tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=levels, encoder_size="tiny", decoder_size="tiny"),
                                  losses=SimpleNamespace(disc_weight=0.4, perceptual_weight=0.0, gram_weight=0.0, perceptual_samples_per_step=24,
                                                         perceptual_sampling_size=128)),
        discriminator=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], model_size="tiny"),
                                      losses=SimpleNamespace(gp_weight=0.1, gp_noise=0.1, centering_weight=0.01)),
        training=SimpleNamespace(main=SimpleNamespace(torch_compile=False, max_steps=1000)))
    model = TiTok(cfg)
    model.load_state_dict(seeded_titok_state(0), strict=True)
    model = model.to(device, dtype).train()
    loss_module = ReconstructionLoss(cfg)
    loss_module.disc_model.load_state_dict(seeded_tower_state("encoder", "tiny", (4, 8, 8), 3, 1, seed=77), strict=True)
    loss_module = loss_module.to(device, dtype).train()
    opt_g = make_optimizer(model)
    opt_d = make_discriminator_optimizer(loss_module) if loss_module is not None else None
    logger = CodebookLogger(4375, world_size=world)

    freeze_python_gc()
    host_threads = limit_host_threads() if os.environ.get("TTV_KEEP_TORCH_THREADS") != "1" else torch.get_num_threads()

    def one_step(batch):
        clips, counts = batch["video"], batch["token_counts"].tolist()
        if loss_module is not None:
            d, idx = gan_training_step(model, loss_module, clips, counts, opt_g, opt_d, overlap=not args.no_overlap)
            loss = d.get("gen/total_loss", d.get("gen/loss", next(iter(d.values()))))
        else:
            from titok_video_amd.train import training_step
            loss, _g, idx = training_step(model, clips, counts, opt_g, overlap=not args.no_overlap)
        logger(torch.split(idx, counts))
        return len(clips), loss

    # loader: worker processes decode + batch, one thread uploads and normalises on the GPU a few batches ahead (titok_video_amd/loader.py)
    it = iter(equal_steps(loader.batches(device, dtype)))
    if args.repeat_first_batch:
        import itertools
        first = next(it)
        it = itertools.repeat(first)
    if args.preload:
        pre = [next(it) for _ in range(args.warmup + args.steps)]
        torch.cuda.synchronize(device)
        it = iter(pre)
        if os.environ.get("TTV_DIAG_WARM_PLANS") == "1":      # one untimed pass over the same batches with an unbounded plan cache: the timed pass finds every plan
            from titok_video_amd import plan as plan_mod0
            plan_mod0._PLAN_CACHE_MAX = 1 << 20
            for b in pre:
                one_step(b)
            torch.cuda.synchronize(device)
    phase = {"loader_wait": 0.0, "get_plan": 0.0, "batch_for": 0.0}
    if args.phase_times:
        from titok_video_amd import plan as plan_mod
        from titok_video_amd.model.base import blocks as blocks_mod

        def timed(fn, key):
            def w(*a, **k):
                t = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    phase[key] += time.perf_counter() - t
            return w
        for k in ("model_fwd", "loss_fwd", "backward", "clip", "opt_step", "logger", "empty_big"):
            phase[k] = 0.0
        model.forward = timed(model.forward, "model_fwd")
        loss_module.forward = timed(loss_module.forward, "loss_fwd")
        torch.Tensor.backward = timed(torch.Tensor.backward, "backward")
        torch.nn.utils.clip_grad_norm_ = timed(torch.nn.utils.clip_grad_norm_, "clip")
        opt_g.step = timed(opt_g.step, "opt_step")
        opt_d.step = timed(opt_d.step, "opt_step")
        for o in (opt_g, opt_d):          # optim.HipAdamW: clip + step in one call (train.clip_and_step)
            if hasattr(o, "clip_and_step"):
                o.clip_and_step = timed(o.clip_and_step, "opt_step")
        real_empty = torch.empty

        def empty(*a, **k):           # allocations above 1 MiB (tapes, workspaces): caching-allocator misses show up here
            t = time.perf_counter()
            out = real_empty(*a, **k)
            if out.numel() * out.element_size() > (1 << 20):
                phase["empty_big"] += time.perf_counter() - t
            return out
        torch.empty = empty
        blocks_mod.get_plan = timed(blocks_mod.get_plan, "get_plan")
        plan_mod.BatchPlan.batch_for = timed(plan_mod.BatchPlan.batch_for, "batch_for")
        real_it = it

        def waited():
            while True:
                t = time.perf_counter()
                b = next(real_it)
                phase["loader_wait"] += time.perf_counter() - t
                yield b
        it = waited()
    for _ in range(args.warmup):
        one_step(next(it))
    for k in phase:
        phase[k] = 0.0
    mem0 = torch.cuda.memory_stats(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    tprof = None
    if args.torch_profile and rank == 0:
        from torch.profiler import ProfilerActivity, profile
        tprof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False, with_stack=False)
        tprof.__enter__()
    prof = None
    if args.host_profile and rank == 0:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    n_clips, loss = 0, None
    for _ in range(args.steps):
        n, loss = one_step(next(it))
        n_clips += n
    if tprof is not None:
        torch.cuda.synchronize(device)
        tprof.__exit__(None, None, None)
        with open(args.torch_profile, "w") as f:
            f.write(tprof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=70))
            f.write("\n\n")
            f.write(tprof.key_averages().table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=70))
    if prof is not None:
        import pstats
        prof.disable()
        with open(args.host_profile, "w") as f:
            st = pstats.Stats(prof, stream=f)
            st.sort_stats("cumulative").print_stats(70)
            st.sort_stats("tottime").print_stats(45)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    tot = torch.tensor([float(n_clips), elapsed], dtype=torch.float64, device="cpu" if (rehearsal or world == 1) else device)
    if world > 1:
        cl = tot[:1].clone()
        dist.all_reduce(cl)
        tm = tot[1:].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        tot = torch.cat([cl, tm])
    # codebook statistics: collective (every rank calls it the same number of times)
    scores = logger.get_scores()
    red = model.__dict__.get("_dp_reducer")
    if rank == 0:
        line = {"metric": "video clips/sec, training step (generator + discriminator), whole node", "value": float(tot[0] / tot[1]), "unit": "clips/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * float(tot[1]) / args.steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic shards",
                "config": {"workload": "BASELINE config #3: configs/tiny.yaml, synthetic tar shards, token budget %d, K ~ U[1,128], reference G + D step "
                                       "(L1 + relativistic GAN, LPIPS off), AdamW, clip 1.0" % args.seq_len,
                           "host_threads": host_threads, "parallelism": f"dp{world}", "backend": args.backend if world > 1 else "none",
                           "grad_allreduce": "after backward" if args.no_overlap else "overlapped with backward (per layer slice, communication stream)",
                           "allreduce_bytes_last_backward": int(red.bytes_reduced) if red is not None else 0,
                           "allreduce_slices_last_backward": int(red.slices) if red is not None else 0},
                "loss_after_steps": float(loss), "codebook": scores}
        if args.phase_times:
            line["host_ms_per_step"] = {k: round(1e3 * v / args.steps, 3) for k, v in phase.items()}
            mem1 = torch.cuda.memory_stats(device)
            line["allocator_per_step"] = {k: (mem1[k] - mem0[k]) / args.steps for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")}
            line["allocator_reserved_MiB"] = mem1["reserved_bytes.all.current"] / 2**20
        print(json.dumps(line), flush=True)
    loader.close()
    if world > 1:
        dist.barrier()
        from titok_video_amd.data import drop_control_groups
        drop_control_groups()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
