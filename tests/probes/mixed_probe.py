"""Diagnostic: the mixed-precision training-step test's two models - token indices and per-parameter gradient cosines."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_hip_backward import config, DEV
from titok_video_amd.model.titok import TiTok
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
from titok_video_amd.train import l1_reconstruction_loss
shapes, counts = [(4, 16, 16), (8, 32, 48), (4, 8, 24)], [2, 5, 3]
clips = synthetic_clips(shapes, seed=31, dtype=torch.bfloat16, device=DEV)
ms = []
for dt in (torch.float32, torch.bfloat16):
    m = TiTok(config()); m.load_state_dict(seeded_titok_state(0), strict=True)
    m = (m.to(DEV) if dt is torch.float32 else m.to(DEV, dt)).train()
    recon, out = m(clips, counts)
    l1_reconstruction_loss(recon, [c * 0.5 for c in clips]).backward()
    ms.append((m, out["indices"].cpu()))
print("indices fp32-master:", ms[0][1].tolist())
print("indices bf16-params:", ms[1][1].tolist())
for (n, p), (_, q) in zip(ms[0][0].named_parameters(), ms[1][0].named_parameters()):
    g, h = p.grad.double().flatten(), q.grad.double().flatten()
    if float(h.norm()) > 0 and p.numel() >= 4096:
        c = float((g @ h) / (g.norm() * h.norm() + 1e-30))
        if c < 0.97:
            print(f"{n}: cos {c:.4f}")
