#!/usr/bin/env python3
"""Headline benchmark: video clips/sec, encode+decode (TiTok.forward), whole node.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): configs/tiny.yaml model, bf16, batch 32 synthetic 16x128x128 RGB clips per GPU,
K = 128 latent tokens per clip; one step = one TiTok.forward (encode -> FSQ -> decode) over the batch.  Inputs are
resident in HBM before the timed region.  N > 1: one process per GPU (launched by torch.distributed.run), every rank
tokenises its own 32 clips, no data-path collective (clips are independent - SURVEY.md 8e), scaling = weak;
the timed region is bracketed by barrier + synchronize and the MAX over ranks is reported.

Two independent batches are kept in flight per GPU by default (--in-flight 2, titok_video_amd/pipeline.py: one HIP stream each;
inference batches do not depend on each other).  Every step is still one full TiTok.forward over 32 clips and exactly K of them
are timed; ms_per_step is the elapsed time / K.  `one_batch_at_a_time` reports the same model with a single chain of launches
(5 untimed-for-the-headline steps after the timed region), and --in-flight 1 times that mode as the headline instead.

Extra objects on the JSON line:
  roofline     : the dominant kernel, timed live with HIP events recorded inside the C library on the
                 stream the kernel is launched on, during the timed steps; achieved = algorithmic FLOPs / launch / time.
                 With two chains in flight a launch shares the CUs with the other chain's kernels, so its bracketed duration
                 is longer than alone: `roofline` carries the launches timed with one chain in flight (a few steps right after
                 the timed region, same hook), `roofline.in_timed_region` what the hook saw during the timed region.
  cpu_baseline : the CPU oracle (a port of the reference algorithm, verified against the reference's own code) timed
                 on this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
  parity       : token-index agreement of the GPU path with that oracle on the sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from titok_video_amd import _lib  # noqa: E402
from titok_video_amd.model.titok import TiTok  # noqa: E402
from titok_video_amd.pipeline import ForwardPipeline  # noqa: E402
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips  # noqa: E402

LEVELS = [7, 5, 5, 5, 5]
CLIP = (16, 128, 128)
BATCH = 32
K_TOKENS = 128
PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16 peak, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0


def tiny_config():
    return SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(
        patch_size=[4, 8, 8], fsq_levels=LEVELS, encoder_size="tiny", decoder_size="tiny")))


def tower_flops_per_clip(S, P, Kt, d=256, layers=4, g=128, inner=704, pd=768, C=5):
    """Algorithmic FLOPs of encode+decode for one clip (SURVEY.md 8d formula; 2*M*N*K per GEMM, 4*S^2*d attention)."""
    per_layer = 2 * S * d * (2 * d + 2 * g) + 4 * S * S * d + 2 * S * d * d + 2 * S * d * 2 * inner + 2 * S * inner * d
    return 2 * layers * per_layer + 2 * 2 * P * pd * d + 2 * 2 * Kt * d * C


def kernel_flops_per_launch(kclass, n_clips, S, d=256, g=128, inner=704):
    L = n_clips * S
    return {
        "attention": n_clips * 4.0 * S * S * d,
        "gemm_qkv": 2.0 * L * d * (2 * d + 2 * g),
        "gemm_geglu": 2.0 * L * d * 2 * inner,
    }[kclass]


def cpu_baseline(sd, n_sample=4, max_runs=5, budget_s=25.0):
    """Oracle timed on the host cores; bounded sample of the workload (n_sample clips per pass)."""
    from oracle import titok_oracle as O
    clips = synthetic_clips([CLIP] * n_sample, seed=1234)
    counts = [K_TOKENS] * n_sample
    # many small ops: more threads than ~32 only adds synchronisation cost; use what is fastest and report it
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    times = []
    t_begin = time.perf_counter()
    out = None
    with torch.no_grad():
        for i in range(max_runs + 1):
            t0 = time.perf_counter()
            out = O.titok_forward(clips, counts, sd, LEVELS)
            dt = time.perf_counter() - t0
            if i > 0:
                times.append(dt)
            if time.perf_counter() - t_begin > budget_s and times:
                break
    times.sort()
    med = times[len(times) // 2]
    return {"value": n_sample / med, "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"{n_sample} clips 16x128x128 K=128 fp32 no_grad, median of {len(times)} passes after 1 warm-up"}, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--kernel", default="attention", choices=["attention", "gemm_qkv", "gemm_geglu"],
                    help="kernel class whose launches are timed for the roofline object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="independent batches in flight per GPU (titok_video_amd.pipeline.ForwardPipeline: one HIP stream each); "
                         "1 = one batch at a time")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.gpus != world and distributed:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal switch for boxes with fewer GPUs than ranks (CPU-side collective, every rank on cuda:0); never set by the driver
    rehearsal = os.environ.get("TTV_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)      # "nccl" is RCCL on ROCm

    sd = seeded_titok_state(0)
    model = TiTok(tiny_config())
    model.load_state_dict(sd, strict=True)
    model = model.to(device, torch.bfloat16).eval()
    clips = synthetic_clips([CLIP] * BATCH, seed=1234 + rank, dtype=torch.bfloat16, device=device)
    counts = [K_TOKENS] * BATCH
    S = K_TOKENS + (CLIP[0] // 4) * (CLIP[1] // 8) * (CLIP[2] // 8)
    P = S - K_TOKENS

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    # A step = TiTok.forward (encode + quantise + decode) of one batch of BATCH clips.  Batches are independent (inference /
    # evaluation: SURVEY.md 8e), so --in-flight of them are kept in flight on separate HIP streams: each is still a chain of 27
    # dependent launches over 32 clips, the hardware fills one chain's partly empty rounds and launch gaps with the other's blocks.
    pipe = ForwardPipeline(model, depth=args.in_flight) if args.in_flight > 1 else None

    def step():
        if pipe is not None:
            pipe.submit(clips, counts)       # outputs are dropped here; a consumer would call pipe.result(ticket)
        else:
            model(clips, counts)

    with torch.no_grad():
        for _ in range(max(args.warmup, args.in_flight)):     # at least one untimed pass per stream (workspace, plan, weight pack)
            step()
        lib = _lib.lib()
        launches_per_step = 8     # 4 layers x 2 towers
        barrier()
        _lib.check(lib.ttv_prof_begin(_lib.KERNEL_CLASSES[args.kernel], launches_per_step * args.steps + 8), "prof_begin")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()                            # torch.cuda.synchronize covers every stream of the device
        elapsed = time.perf_counter() - t0
        tot_ms, cnt = C.c_double(0), C.c_int(0)
        _lib.check(lib.ttv_prof_end(C.byref(tot_ms), C.byref(cnt)), "prof_end")
        # a few steps one batch at a time, outside the timed region: the same kernels without a neighbour on the part
        seq_steps = 5
        iso_ms, iso_cnt, seq_elapsed = C.c_double(0), C.c_int(0), 0.0
        if pipe is not None and rank == 0:
            _lib.check(lib.ttv_prof_begin(_lib.KERNEL_CLASSES[args.kernel], launches_per_step * seq_steps + 8), "prof_begin")
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for _ in range(seq_steps):
                model(clips, counts)
            torch.cuda.synchronize(device)
            seq_elapsed = time.perf_counter() - t1
            _lib.check(lib.ttv_prof_end(C.byref(iso_ms), C.byref(iso_cnt)), "prof_end")

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        clips_total = world * BATCH * args.steps
        value = clips_total / elapsed
        kern_ms = tot_ms.value / max(cnt.value, 1)
        kflops = kernel_flops_per_launch(args.kernel, BATCH, S)
        achieved = kflops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        flops_clip = tower_flops_per_clip(S, P, K_TOKENS)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.kernel)
            except Exception:
                traffic = None
        line = {
            "metric": "video clips/sec encode+decode (whole node)", "value": value, "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs/tiny.yaml (tiny enc+dec, FSQ [7,5,5,5,5]) bf16, batch 32 x 16x128x128 clips per GPU, "
                                   "K=128 latent tokens, TiTok.forward encode+decode", "clips_per_gpu": BATCH,
                       "tokens_per_clip": S, "parallelism": f"dp{world} (clips sharded, no collective)",
                       "batches_in_flight": args.in_flight,
                       "algorithmic_gflop_per_clip": flops_clip / 1e9,
                       "whole_path_mfma_frac": value / world * flops_clip / (PEAK_BF16_TFLOPS * 1e12)},
            "roofline": {"kernel": args.kernel, "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "avg_launch_ms": kern_ms, "launches_timed": cnt.value, "flops_per_launch": kflops},
        }
        if pipe is not None and iso_cnt.value > 0:
            # With two chains in flight a launch shares the CUs with the other chain's kernels, so its event-bracketed duration is
            # longer than the same launch alone and says little about the kernel.  `roofline` therefore carries the launches timed
            # with ONE chain in flight (seq_steps steps right after the timed region, same HIP-event hook); what the hook saw during
            # the timed region itself is kept under `in_timed_region`.
            iso = iso_ms.value / iso_cnt.value
            overlapped = dict(line["roofline"])
            line["roofline"].update({"achieved": kflops / (iso * 1e-3) / 1e12, "frac": kflops / (iso * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                                     "avg_launch_ms": iso, "launches_timed": iso_cnt.value,
                                     "measured": f"{seq_steps} steps with one batch in flight, right after the timed region",
                                     "in_timed_region": {"concurrent_chains": args.in_flight, "avg_launch_ms": overlapped["avg_launch_ms"],
                                                         "achieved": overlapped["achieved"], "frac": overlapped["frac"],
                                                         "launches_timed": overlapped["launches_timed"]}})
            line["one_batch_at_a_time"] = {"value": BATCH * seq_steps / seq_elapsed, "ms_per_step": 1e3 * seq_elapsed / seq_steps,
                                           "steps": seq_steps, "n_gpus": 1}
        if world == 1 and not args.no_cpu_baseline:
            base, ref = cpu_baseline(sd)
            line["cpu_baseline"] = base
            from oracle import titok_oracle as O
            with torch.no_grad():
                c4 = synthetic_clips([CLIP] * 4, seed=1234, dtype=torch.bfloat16, device=device)
                _, o4 = model(c4, [K_TOKENS] * 4)
            idx = o4["indices"].cpu()
            ref_idx, margin = ref[1], O.fsq_margin(ref[3])
            safe = margin > 0.08
            line["parity"] = {"index_match_raw": float((idx == ref_idx).float().mean()),
                              "index_match_margin_gt_0.08": float((idx[safe] == ref_idx[safe]).float().mean()),
                              "tokens": int(idx.numel()), "safe_tokens": int(safe.sum()), "oracle": "fp32 CPU"}
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
