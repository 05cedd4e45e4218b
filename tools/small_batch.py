#!/usr/bin/env python3
"""Per-kernel time of one tower forward at a small packed batch (the reference trains under a 6144-token budget,
configs/tiny.yaml:65): where the latency-bound launches are.  GPU box only."""
import os, sys, torch
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd.model.titok import TiTok
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
from titok_video_amd import _lib
import ctypes as C
B = int(os.environ.get("B", "5"))
cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny")))
m = TiTok(cfg); m.load_state_dict(seeded_titok_state(0)); m = m.to("cuda:0", torch.bfloat16).eval()
clips = synthetic_clips([(16, 128, 128)] * B, seed=1, dtype=torch.bfloat16, device="cuda:0")
counts = [128] * B
lib = _lib.lib()
with torch.no_grad():
    for _ in range(5): m(clips, counts)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(50): m(clips, counts)
    torch.cuda.synchronize()
    print(f"B={B} rows={B*1152}: {(time.perf_counter()-t0)/50*1e3:.3f} ms per forward")
    for name, cls, per in (("attention", 1, 8), ("gemm_qkv", 2, 8), ("layer_tail", 3, 8)):
        lib.ttv_prof_begin(cls, 400)
        for _ in range(20): m(clips, counts)
        tot, cnt = C.c_double(0), C.c_int(0)
        lib.ttv_prof_end(C.byref(tot), C.byref(cnt))
        print(f"  {name:12s} {tot.value/cnt.value*1e3:7.1f} us x {per}")
