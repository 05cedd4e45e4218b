"""Golden fixture for BASELINE config #4 (base towers, one 32x256x256 clip, K = 1024 latent tokens: S = 9216 rows, 144 key tiles,
72 query blocks per head) from the REFERENCE's own modules - same method as make_golden.py (stand-ins for flash_attn / xformers
stating their published definitions; the reference's TiTok imported unmodified).  fp32 run + the reference's own bf16 run (the
yardstick of the HIP bf16 path).  Build container only:   python tests/golden/make_golden_base.py      (several minutes on 8 cores)

The reference ships no base config: dims come from its get_model_dims('base') (model/base/utils.py:8-23); FSQ levels
[8,8,8,6,5] are the "16k" set its configs/tiny.yaml:17 comment names; weights from the seed recipe with gain 3 (non-degenerate indices).
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (puts the repo root and /root/reference on sys.path)

from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips  # noqa: E402

LEVELS = [8, 8, 8, 6, 5]
SHAPE, K, GAIN, CLIP_SEED = (32, 256, 256), 1024, 3.0, 4044


def main():
    torch.manual_seed(0)
    MG.install_standins()
    from model.titok import TiTok
    from types import SimpleNamespace
    cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=LEVELS, encoder_size="base", decoder_size="base")))
    sd = seeded_titok_state(0, "base", "base", gain=GAIN)
    out = {"shape": np.array(SHAPE, dtype=np.int32), "count": np.int32(K), "clip_seed": np.int32(CLIP_SEED), "weight_seed": np.int32(0),
           "weight_gain": np.float32(GAIN), "levels": np.array(LEVELS, dtype=np.int32)}
    codes32 = None
    for dtype, tag in ((torch.float32, ""), (torch.bfloat16, "_refbf16")):
        model = TiTok(cfg).eval()
        model.load_state_dict(sd, strict=True)
        model = model.to(dtype)
        clips = synthetic_clips([SHAPE], seed=CLIP_SEED, dtype=dtype)
        tc = torch.tensor([K], dtype=torch.int32)
        grids = torch.tensor([SHAPE], dtype=torch.int32)
        t0 = time.time()
        with torch.no_grad():
            z = model.encoder(clips, tc, grids)
            codes, d = model.quantize(z)
            bounded = model.quantize.bound(z.float())
            if codes32 is None:
                codes32 = codes
            recon = model.decode(codes32.to(dtype), tc, grids)[0]          # decoder on the fp32 run's codes in both runs
        idx = d["indices"]
        print(f"{dtype}: {time.time() - t0:.0f} s; distinct indices {len(set(idx.tolist()))}/{idx.numel()}; recon std {recon.float().std():.3f}", flush=True)
        out["indices" + tag] = idx.numpy().astype(np.int32)
        out["bounded" + tag] = bounded.float().numpy()
        out["recon_sample" + tag] = recon.float()[:, ::4, ::8, ::8].contiguous().numpy()      # [3, 8, 32, 32]
        if tag == "":
            out["z"] = z.float().numpy()
            out["recon_std"] = np.float64(recon.double().std().item())
    np.savez_compressed(os.path.join(HERE, "titok_base_cfg4.npz"), **out)
    print("wrote titok_base_cfg4.npz")


if __name__ == "__main__":
    main()
