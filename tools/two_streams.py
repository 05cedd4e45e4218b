#!/usr/bin/env python3
"""Does running two half-batches on two HIP streams fill the idle CUs (tail effects, 192-CU tail kernel, launch gaps)?"""
import os, sys, time, torch
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd.model.titok import TiTok
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny")))
m = TiTok(cfg); m.load_state_dict(seeded_titok_state(0)); m = m.to("cuda:0", torch.bfloat16).eval()
B = 32
clips = synthetic_clips([(16, 128, 128)] * B, seed=1234, dtype=torch.bfloat16, device="cuda:0")
counts = [128] * B
def run_single():
    return m(clips, counts)
def make_split(nsplit):
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    per = B // nsplit
    def run():
        cur = torch.cuda.current_stream()
        outs = []
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(m(clips[i * per:(i + 1) * per], counts[i * per:(i + 1) * per]))
        for s in streams:
            cur.wait_stream(s)
        return outs
    return run
def bench(fn, n=50):
    with torch.no_grad():
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
t1 = bench(run_single); print(f"single stream : {t1*1e3:.3f} ms  {B/t1:8.0f} clips/s")
for k in (2, 4):
    t = bench(make_split(k)); print(f"{k} streams     : {t*1e3:.3f} ms  {B/t:8.0f} clips/s")

# ---- two FULL batches in flight (consecutive steps on two streams, one model instance per stream: the workspace is per model)
m2 = TiTok(cfg); m2.load_state_dict(seeded_titok_state(0)); m2 = m2.to("cuda:0", torch.bfloat16).eval()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def run_pair():
    with torch.cuda.stream(sa):
        a = m(clips, counts)
    with torch.cuda.stream(sb):
        b = m2(clips, counts)
    return a, b
with torch.no_grad():
    for _ in range(10): run_pair()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): run_pair()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
print(f"2 full batches in flight: {t*1e3:.3f} ms per pair  {2*B/t:8.0f} clips/s")
