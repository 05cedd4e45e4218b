import os, sys, torch
sys.path.insert(0, "/root/repo")
from types import SimpleNamespace
from oracle import titok_oracle as O
from titok_video_amd.model.titok import TiTok
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips
size = os.environ.get("SIZE", "small")
LEVELS=[7,5,5,5,5]
cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4,8,8], fsq_levels=LEVELS, encoder_size=size, decoder_size=size)))
sd = seeded_titok_state(3, encoder_size=size, decoder_size=size, gain=float(os.environ.get("GAIN", "3.0")))
shapes, counts = [(4, 16, 16), (8, 16, 24), (4, 32, 16)], [128, 128, 128]
clips_cpu = synthetic_clips(shapes, seed=13)
with torch.no_grad():
    _r, ref_idx, _z, ref_b = O.titok_forward(clips_cpu, counts, sd, LEVELS, size, size)
    _r, y_idx, _z, y_b = O.titok_forward([c.to(torch.bfloat16) for c in clips_cpu], counts, sd, LEVELS, size, size)
m = TiTok(cfg); m.load_state_dict(sd, strict=True); m = m.to("cuda:0", torch.bfloat16).eval()
with torch.no_grad():
    m.encode([c.to("cuda:0", torch.bfloat16) for c in clips_cpu], counts, want_bounded=True)
err = (m.last_bounded.float().cpu() - ref_b).abs(); yerr = (y_b.float() - ref_b).abs()
print(os.environ.get("TAG",""), f"HIP mean {float(err.mean()):.5f} max {float(err.max()):.4f} | yard mean {float(yerr.mean()):.5f} max {float(yerr.max()):.4f} | ratio {float(err.mean()/yerr.mean()):.3f}")
