#!/usr/bin/env python3
"""BASELINE.json config #4: base-size towers (d=768, 12 layers, heads 12/4), 32x256x256 clips, K=1024, bf16, 1 GPU.
The reference ships no base config; dims come from get_model_dims('base').  FSQ levels [8,8,8,6,5]."""
import json, os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd.model.titok import TiTok
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
B = int(os.environ.get("B", "4"))
cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=[8, 8, 8, 6, 5], encoder_size="base", decoder_size="base")))
m = TiTok(cfg); m.load_state_dict(seeded_titok_state(0, "base", "base", gain=2.0)); m = m.to("cuda:0", torch.bfloat16).eval()
clips = synthetic_clips([(32, 256, 256)] * B, seed=1, dtype=torch.bfloat16, device="cuda:0")
counts = [1024] * B
with torch.no_grad():
    for _ in range(2):
        m(clips, counts)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        recon, out = m(clips, counts)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
S, P, d, layers, g, I, pd = 9216, 8192, 768, 12, 256, 2048, 768
per_layer = 2 * S * d * (2 * d + 2 * g) + 4 * S * S * d + 2 * S * d * d + 2 * S * d * 2 * I + 2 * S * I * d
flops = 2 * layers * per_layer + 4 * P * pd * d
print(json.dumps({"config": "base 32x256x256 K=1024 bf16", "batch": B, "ms_per_step": 1e3 * dt, "clips_per_s": B / dt,
                  "tflops_per_clip": flops / 1e12, "achieved_tflops": B * flops / dt / 1e12, "mfma_frac": B * flops / dt / 2.5e15,
                  "distinct_indices": int(out["indices"].unique().numel())}))
