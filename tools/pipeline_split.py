#!/usr/bin/env python3
"""Batches in flight x batch size at a fixed number of clips in flight: ForwardPipeline(depth) fed with batches of 32 / depth' clips.
Does finer interleaving (4 x 16 clips) fill the part better than the benchmark's 2 x 32?  Prints clips/s per configuration."""
import os, sys, time, torch
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd.model.titok import TiTok
from titok_video_amd.pipeline import ForwardPipeline
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny")))
m = TiTok(cfg); m.load_state_dict(seeded_titok_state(0)); m = m.to("cuda:0", torch.bfloat16).eval()
N = 64
clips = synthetic_clips([(16, 128, 128)] * N, seed=1234, dtype=torch.bfloat16, device="cuda:0")
def run(depth, bs, steps=240):
    pipe = ForwardPipeline(m, depth=depth)
    batches = [(clips[i:i + bs], [128] * bs) for i in range(0, N, bs)]
    def go(n):
        tickets = []
        for i in range(n):
            c, k = batches[i % len(batches)]
            tickets.append(pipe.submit(c, k))
            if len(tickets) > depth:
                pipe.result(tickets.pop(0))
        for t in tickets:
            pipe.result(t)
        pipe.drain()
    with torch.no_grad():
        go(4 * depth); torch.cuda.synchronize()
        n = steps * 32 // bs
        t0 = time.perf_counter(); go(n); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"depth {depth} x {bs:2d} clips: {n * bs / dt:9.0f} clips/s", flush=True)
for depth, bs in ((1, 32), (2, 32), (2, 16), (4, 16), (3, 32), (4, 8), (8, 8), (2, 64)):
    run(depth, bs)
