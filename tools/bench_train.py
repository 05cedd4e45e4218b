#!/usr/bin/env python3
"""Training-step timing at the benchmark shape (32 x 16x128x128 clips, K=128, bf16): forward (tape) + L1 + backward + AdamW.
B= batch (5 = the reference's 6144-token budget), GRAPH=1: the whole step captured once in a HIP graph (torch.cuda.CUDAGraph) and
replayed - the step is a fixed launch sequence for a fixed batch shape, and at small batches the host cannot issue ~280 launches
as fast as the GPU runs them."""
import json, os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd.model.titok import TiTok
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
from titok_video_amd.train import freeze_python_gc, limit_host_threads, make_optimizer, training_step
B = int(os.environ.get("B", "32"))
if os.environ.get("TTV_DEBUG"):   # diagnostics bits of ttv_debug_set (A/B runs on one box)
    from titok_video_amd import _lib
    _lib.lib().ttv_debug_set(int(os.environ["TTV_DEBUG"]))
cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny")))
m = TiTok(cfg); m.load_state_dict(seeded_titok_state(0)); m = m.to("cuda:0", torch.bfloat16).train()
clips = synthetic_clips([(16, 128, 128)] * B, seed=1, dtype=torch.bfloat16, device="cuda:0")
counts = [128] * B
GRAPH = os.environ.get("GRAPH", "0") == "1"
opt = make_optimizer(m, capturable=True) if GRAPH else make_optimizer(m)
if os.environ.get("GC_FREEZE", "1") == "1":
    freeze_python_gc()
    limit_host_threads()
n = int(os.environ.get("STEPS", "10"))
if GRAPH:
    from titok_video_amd.train import GraphedTrainingStep
    step = GraphedTrainingStep(m, opt, clips, counts)
    for _ in range(3):
        loss, gn, _ = step(clips)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        loss, gn, _ = step(clips)
    torch.cuda.synchronize()
else:
    for _ in range(3):
        training_step(m, clips, counts, opt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        loss, gn, _ = training_step(m, clips, counts, opt)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(json.dumps({"train_ms_per_step": 1e3 * dt, "clips_per_s": B / dt, "batch": B, "loss": float(loss), "graph": GRAPH}))
