#!/usr/bin/env python3
"""The gate columns formed inside the attention launch (TTV_ATTN_GATE_X=1, default) against the stored columns (=0): token indices and
reconstruction of the benchmark forward, one process per setting.  GPU box only."""
import os
import subprocess
import sys

CODE = r'''
import os, sys, torch
sys.path.insert(0, ".")
import bench
wl = bench.WORKLOADS["tiny"]
dev = torch.device("cuda:0")
model, sd = bench.build_model(wl, "fsq", dev, torch.bfloat16)[:2]
from titok_video_amd.synthetic import synthetic_clips
clips = synthetic_clips([wl["clip"]] * 32, seed=wl["clip_seed"], dtype=torch.bfloat16, device=dev)
with torch.no_grad():
    recon, out = model(clips, [wl["k_tokens"]] * 32)
idx = out["indices"].cpu()
r = torch.cat([x.float().flatten() for x in recon]).cpu()
torch.save({"idx": idx, "recon": r}, sys.argv[1])
'''
outs = []
for v in ("1", "0"):
    env = dict(os.environ, TTV_ATTN_GATE_X=v)
    path = f"/tmp/gate_x_{v}.pt"
    p = subprocess.run([sys.executable, "-c", CODE, path], env=env, capture_output=True, text=True)
    if p.returncode:
        print(p.stderr[-2000:])
        sys.exit(1)
    import torch
    outs.append(torch.load(path))
a, b = outs
same = float((a["idx"] == b["idx"]).float().mean())
d = (a["recon"] - b["recon"]).abs()
print(f"indices equal: {same:.4f} of {a['idx'].numel()}; recon max abs diff {float(d.max()):.4e}, mean {float(d.mean()):.4e}, recon scale {float(b['recon'].abs().mean()):.3f}")
