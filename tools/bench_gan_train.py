#!/usr/bin/env python3
"""Full reference training step (generator + discriminator, train.py:64-107) at the benchmark batch: tiny towers, 32 clips of
16x128x128, K=128, bf16; perceptual terms off.  8 tower forwards + 5 tower backwards per step.  GPU box only."""
import json, os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd.model.titok import TiTok
from titok_video_amd.model.losses import ReconstructionLoss
from titok_video_amd.synthetic import seeded_titok_state, seeded_tower_state, synthetic_clips
from titok_video_amd.train import freeze_python_gc, limit_host_threads, gan_training_step, make_discriminator_optimizer, make_optimizer
B = int(os.environ.get("B", "32"))
cfg = SimpleNamespace(
    tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny"),
                              losses=SimpleNamespace(disc_weight=0.4, perceptual_weight=0.0, gram_weight=0.0, perceptual_samples_per_step=24,
                                                     perceptual_sampling_size=128)),
    discriminator=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], model_size="tiny"),
                                  losses=SimpleNamespace(gp_weight=0.1, gp_noise=0.1, centering_weight=0.01)),
    training=SimpleNamespace(main=SimpleNamespace(torch_compile=False, max_steps=1000)))
m = TiTok(cfg); m.load_state_dict(seeded_titok_state(0)); m = m.to("cuda:0", torch.bfloat16).train()
lm = ReconstructionLoss(cfg); lm.disc_model.load_state_dict(seeded_tower_state("encoder", "tiny", (4, 8, 8), 3, 1, seed=77))
lm = lm.to("cuda:0", torch.bfloat16).train()
clips = synthetic_clips([(16, 128, 128)] * B, seed=1, dtype=torch.bfloat16, device="cuda:0")
counts = [128] * B
og, od = make_optimizer(m), make_discriminator_optimizer(lm)
freeze_python_gc()
limit_host_threads()
for _ in range(3):
    gan_training_step(m, lm, clips, counts, og, od)
torch.cuda.synchronize()
n = int(os.environ.get("STEPS", "10"))
t0 = time.perf_counter()
for _ in range(n):
    ld, _ = gan_training_step(m, lm, clips, counts, og, od)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(json.dumps({"gan_train_ms_per_step": 1e3 * dt, "clips_per_s": B / dt, "batch": B, "gen_total": float(ld["gen/total_loss"]),
                  "disc_total": float(ld["disc/total_loss"])}))
