"""Golden fixture for the model sizes beyond tiny (reference get_model_dims: small / base / large, model/base/utils.py:8-23) from the
REFERENCE's own modules - same method as make_golden.py (stand-ins for flash_attn / xformers stating their published definitions; the
reference's TiTok imported unmodified): its fp32 run and ITS OWN bf16 run on three small clips with K = 128 latent tokens each
(384 tokens per size).  The bf16 run is the yardstick of tests/test_hip_parity.py::test_other_model_sizes_match_oracle - round 2 used
the CPU oracle evaluated in bf16 there, which is a more accurate bf16 execution than the reference's (fewer roundings), so the HIP
path read 25 % "worse" than a yardstick the reference itself does not meet.  Build container only:
    python tests/golden/make_golden_sizes.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (puts the repo root and /root/reference on sys.path)

from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips  # noqa: E402

LEVELS = [7, 5, 5, 5, 5]
SHAPES, COUNTS, CLIP_SEED, WEIGHT_SEED, GAIN = [(4, 16, 16), (8, 16, 24), (4, 32, 16)], [128, 128, 128], 13, 3, 3.0


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    MG.install_standins()
    from model.titok import TiTok
    out = {"shapes": np.array(SHAPES, dtype=np.int32), "counts": np.array(COUNTS, dtype=np.int32), "clip_seed": np.int32(CLIP_SEED),
           "weight_seed": np.int32(WEIGHT_SEED), "weight_gain": np.float32(GAIN), "levels": np.array(LEVELS, dtype=np.int32)}
    for size in ("small", "base", "large"):
        cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=LEVELS, encoder_size=size, decoder_size=size)))
        sd = seeded_titok_state(WEIGHT_SEED, size, size, gain=GAIN)
        for dtype, tag in ((torch.float32, ""), (torch.bfloat16, "_refbf16")):
            model = TiTok(cfg).eval()
            model.load_state_dict(sd, strict=True)
            model = model.to(dtype)
            clips = synthetic_clips(SHAPES, seed=CLIP_SEED, dtype=dtype)
            tc = torch.tensor(COUNTS, dtype=torch.int32)
            grids = torch.tensor(SHAPES, dtype=torch.int32)
            with torch.no_grad():
                z = model.encoder(clips, tc, grids)
                _codes, d = model.quantize(z)
                bounded = model.quantize.bound(z.float())
            out[f"{size}_indices{tag}"] = d["indices"].numpy().astype(np.int32)
            out[f"{size}_bounded{tag}"] = bounded.float().numpy()
            print(size, dtype, "distinct indices", len(set(d["indices"].tolist())), flush=True)
        e = np.abs(out[f"{size}_bounded_refbf16"] - out[f"{size}_bounded"])
        print(f"{size}: reference bf16 vs fp32: mean |bounded err| {e.mean():.5f}, max {e.max():.4f}, index mismatches "
              f"{int((out[f'{size}_indices_refbf16'] != out[f'{size}_indices']).sum())}/384")
    np.savez_compressed(os.path.join(HERE, "titok_sizes.npz"), **out)
    print("wrote titok_sizes.npz", os.path.getsize(os.path.join(HERE, "titok_sizes.npz")) // 1024, "KiB")

    # the corners of the loader's sampling ranges (configs/tiny.yaml:57-62) through the tiny model: largest grid with K = 128 and K = 1,
    # smallest grid with K = 1, two small full-K clips - 386 tokens; yardstick of test_sampling_range_extremes_bf16_close_to_oracle
    shapes, counts, seed = [(16, 168, 168), (8, 128, 128), (16, 168, 168), (4, 16, 16), (8, 16, 24)], [128, 1, 1, 128, 128], 77
    ex = {"shapes": np.array(shapes, dtype=np.int32), "counts": np.array(counts, dtype=np.int32), "clip_seed": np.int32(seed),
          "weight_seed": np.int32(0), "levels": np.array(LEVELS, dtype=np.int32)}
    cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=LEVELS, encoder_size="tiny", decoder_size="tiny")))
    sd = seeded_titok_state(0)
    for dtype, tag in ((torch.float32, ""), (torch.bfloat16, "_refbf16")):
        model = TiTok(cfg).eval()
        model.load_state_dict(sd, strict=True)
        model = model.to(dtype)
        clips = [c.to(dtype) for c in synthetic_clips(shapes, seed=seed)]      # as the GPU test: fp32 clips rounded to the run's dtype
        tc = torch.tensor(counts, dtype=torch.int32)
        grids = torch.tensor(shapes, dtype=torch.int32)
        with torch.no_grad():
            z = model.encoder(clips, tc, grids)
            _codes, d = model.quantize(z)
            bounded = model.quantize.bound(z.float())
        ex["indices" + tag] = d["indices"].numpy().astype(np.int32)
        ex["bounded" + tag] = bounded.float().numpy()
    e = np.abs(ex["bounded_refbf16"] - ex["bounded"])
    print(f"extremes: reference bf16 vs fp32: mean |bounded err| {e.mean():.5f}, max {e.max():.4f}, mismatches {int((ex['indices_refbf16'] != ex['indices']).sum())}/386")
    np.savez_compressed(os.path.join(HERE, "titok_extremes.npz"), **ex)
    print("wrote titok_extremes.npz")


if __name__ == "__main__":
    main()
