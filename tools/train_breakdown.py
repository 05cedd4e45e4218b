#!/usr/bin/env python3
"""Where a training step's wall time goes (synchronised sections; GPU box only)."""
import os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd.model.titok import TiTok
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
from titok_video_amd.train import make_optimizer, l1_reconstruction_loss
from titok_video_amd import dp
B = 32
cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny")))
m = TiTok(cfg); m.load_state_dict(seeded_titok_state(0)); m = m.to("cuda:0", torch.bfloat16).train()
clips = synthetic_clips([(16, 128, 128)] * B, seed=1, dtype=torch.bfloat16, device="cuda:0")
counts = [128] * B
opt = make_optimizer(m)
acc = {}
import gc
if os.environ.get("NOGC"): gc.disable()
def sec(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    a = acc.setdefault(name, [0.0, 0.0]); a[0] += t1 - t0; a[1] += t2 - t0
    return r
for it in range(8):
    if it == 3: acc.clear()
    sec("zero_grad", lambda: opt.zero_grad(set_to_none=True))
    recon, out = sec("forward", lambda: m(clips, counts))
    loss = sec("loss", lambda: l1_reconstruction_loss(recon, clips))
    sec("backward", lambda: loss.backward())
    params = [p for p in m.parameters() if p.grad is not None]
    sec("allreduce", lambda: dp.allreduce_mean_by_count([p.grad for p in params], B))
    sec("tiny alloc", lambda: torch.zeros(8, device="cuda:0"))
    sec("grad[0].sum", lambda: params[0].grad.sum())
    sec("foreach_norm", lambda: torch._foreach_norm([p.grad for p in params]))
    sec("clip", lambda: torch.nn.utils.clip_grad_norm_(params, 1.0))
    sec("clip(again)", lambda: torch.nn.utils.clip_grad_norm_(params, 1.0))
    sec("opt.step", lambda: opt.step())
n = 5
for k, (cpu, tot) in acc.items():
    print(f"{k:10s} cpu-issue {1e3*cpu/n:7.2f} ms   wall(with sync) {1e3*tot/n:7.2f} ms")
