#!/usr/bin/env python3
"""Does a two-chain forward get faster while it runs?  ForwardPipeline(depth 2) at the benchmark batch: W warm-up steps, device
synchronise, then 60 steps with an event per step; prints the completion time of every 5th step (ms since the first submit) and
the clips/s of consecutive 10-step windows.  W = 5 and W = 50, twice each."""
import os, sys, time, torch
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd.model.titok import TiTok
from titok_video_amd.pipeline import ForwardPipeline
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny")))
m = TiTok(cfg); m.load_state_dict(seeded_titok_state(0)); m = m.to("cuda:0", torch.bfloat16).eval()
sets = [synthetic_clips([(16, 128, 128)] * 32, seed=s, dtype=torch.bfloat16, device="cuda:0") for s in (1, 2)]
counts = [128] * 32
pipe = ForwardPipeline(m, depth=2)
def run(W, K=60):
    with torch.no_grad():
        for i in range(W):
            pipe.submit(sets[i % 2], counts)
        torch.cuda.synchronize()
        start = torch.cuda.Event(enable_timing=True); start.record()
        evs = []
        for i in range(K):
            _, done = pipe.submit(sets[i % 2], counts)
            e = torch.cuda.Event(enable_timing=True)
            e.record(pipe.streams[i % 2])
            evs.append(e)
        torch.cuda.synchronize()
    t = [start.elapsed_time(e) for e in evs]
    wins = [32 * 10 / ((max(t[i:i + 10]) - (max(t[i - 10:i]) if i else 0.0)) * 1e-3) for i in range(0, K, 10)]
    print(f"W={W:3d}: step completion ms " + " ".join(f"{t[i]:.2f}" for i in range(4, K, 5)))
    print(f"        clips/s per 10-step window: " + " ".join(f"{w:.0f}" for w in wins), flush=True)
for W in (5, 50, 5, 50):
    time.sleep(0.5)
    run(W)
