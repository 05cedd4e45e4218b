import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_hip_parity import build, DEV
from titok_video_amd.synthetic import synthetic_clips
from titok_video_amd import _lib
from oracle import titok_oracle as O
from titok_video_amd.synthetic import seeded_titok_state
model = build(torch.bfloat16)
shapes = [(8, 32, 48), (4, 16, 16), (16, 64, 32), (16, 128, 128)]; counts = [5, 1, 17, 128]
clips = synthetic_clips(shapes, seed=29, dtype=torch.bfloat16, device=DEV)
lib = _lib.lib(); outs = []
for bits in (0, 1 << 15, 1 << 17):
    lib.ttv_debug_set(bits)
    with torch.no_grad():
        model.encode(clips, counts, want_bounded=True)
    torch.cuda.synchronize()
    outs.append(model.last_bounded.float().cpu().clone())
lib.ttv_debug_set(0)
sd = seeded_titok_state(0)
with torch.no_grad():
    _r, ref_idx, _z, ref_b = O.titok_forward([c.float().cpu() for c in clips], counts, sd, [7, 5, 5, 5, 5])
for n, o in zip(("k_qkv256", "k_gemm_k256", "k_qkv256ws"), outs):
    e = (o - ref_b).abs()
    print(f"{n:12s} vs fp32 oracle: mean {e.mean():.5f} max {e.max():.4f}")
d = (outs[0] - outs[1]).abs()
print("k_qkv256 vs k_gemm_k256: mean", float(d.mean()), "max", float(d.max()), "frac > 0.02", float((d > 0.02).float().mean()))
print("k_qkv256 vs ws equal:", torch.equal(outs[0], outs[2]))
