#!/usr/bin/env python3
"""Secondary measurements (not the headline metric): ragged token-budget batches as the reference's loader emits them
(configs/tiny.yaml sampling ranges, train_seq_len 6144) through inference and through a full training step."""
import itertools
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd import plan as plan_mod  # noqa: E402
from titok_video_amd.data import SyntheticClipStream, dynamic_batches  # noqa: E402
from titok_video_amd.model.titok import TiTok  # noqa: E402
from titok_video_amd.synthetic import seeded_titok_state  # noqa: E402
from titok_video_amd.train import make_optimizer, training_step  # noqa: E402

DEV = torch.device("cuda:0")
cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny")))
model = TiTok(cfg)
model.load_state_dict(seeded_titok_state(0))
model = model.to(DEV, torch.bfloat16)
stream = SyntheticClipStream(dtype=torch.bfloat16, device=DEV, seed=1, length=400)
batches = list(itertools.islice(dynamic_batches(stream, (4, 8, 8), (1, 128), 6144, seed=2, max_grid=(16, 168, 168)), 40))
for b in batches:
    b["counts"] = b["token_counts"].tolist()
nclips = sum(len(b["video"]) for b in batches)
rows = sum(sum((v.shape[1] // 4) * (v.shape[2] // 8) * (v.shape[3] // 8) for v in b["video"]) + sum(b["counts"]) for b in batches)

model.eval()
with torch.no_grad():
    for b in batches[:5]:
        model(b["video"], b["counts"])
    torch.cuda.synchronize()
    plan_mod._plan_cache.clear()        # every timed batch builds its plan (a loader never repeats a batch shape)
    t0 = time.perf_counter()
    for b in batches:
        model(b["video"], b["counts"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
out = {"ragged_inference": {"batches": len(batches), "clips": nclips, "packed_rows": rows, "clips_per_s": nclips / dt, "rows_per_s": rows / dt,
                            "ms_per_batch": 1e3 * dt / len(batches)}}

# the same batches with several of them in flight (titok_video_amd.pipeline): small ragged batches under-fill the part even more
from titok_video_amd.pipeline import ForwardPipeline  # noqa: E402
for depth in (2, 3):
    pipe = ForwardPipeline(model, depth=depth)
    for b in batches[:5]:
        pipe.submit(b["video"], b["counts"])
    torch.cuda.synchronize()
    plan_mod._plan_cache.clear()
    t0 = time.perf_counter()
    for b in batches:
        pipe.submit(b["video"], b["counts"])
    torch.cuda.synchronize()
    dtp = time.perf_counter() - t0
    out[f"ragged_inference_{depth}_in_flight"] = {"clips_per_s": nclips / dtp, "ms_per_batch": 1e3 * dtp / len(batches)}

# bf16-mixed as the reference trains (configs/tiny.yaml:70): fp32 master weights, bf16 compute through the clips' dtype
# (the weight packs hold bf16 copies; gradients arrive in fp32 - tests/test_hip_backward.py::test_mixed_precision_...)
model = model.float().train()
opt = make_optimizer(model)
for b in batches[:3]:
    training_step(model, b["video"], b["counts"], opt)
torch.cuda.synchronize()
t0 = time.perf_counter()
nb = 10
for b in batches[:nb]:
    loss, gn, _ = training_step(model, b["video"], b["counts"], opt)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
nc = sum(len(b["video"]) for b in batches[:nb])
out["training_step_bf16_mixed"] = {"steps": nb, "ms_per_step": 1e3 * dt / nb, "clips_per_s": nc / dt, "last_loss": float(loss), "last_grad_norm": float(gn)}
print(json.dumps(out))
