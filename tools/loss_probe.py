#!/usr/bin/env python3
"""Where the bf16 R1 / R2 penalties of the discriminator step pick up their excess over the reference: the fixture's
step through the loss module, printed next to the reference's fp32 and bf16 values.  Run under different switches (TTV_ATTN_THR=0:
exact running maximum per row - a row's softmax then does not depend on which rows share its wave).  GPU box only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_hip_loss as T  # noqa: E402

d, mod, sd, target, recon, noise, to = T.fixture(torch.bfloat16)
tot, parts = mod(to(target), to(recon), disc_forward=True, gp_noise_tensors=to(noise))     # training path (tape forward), as the test
def show(tag, tot, parts):
    print(os.environ.get("TAG", ""), tag, f"total {float(tot):.3f} (ref fp32 {float(d['disc_total']):.3f}, ref bf16 {float(d['disc_total_bf16']):.3f});",
          f"r1 {float(parts['disc/r1_penalty']):.5f} (ref {float(d['disc_r1_penalty']):.5f}); r2 {float(parts['disc/r2_penalty']):.5f} (ref {float(d['disc_r2_penalty']):.5f})")


show("training path (tape forward):", tot.detach(), parts)
with torch.no_grad():       # the same loss through the inference towers (fused kernels)
    tot2, parts2 = mod(to(target), to(recon), disc_forward=True, gp_noise_tensors=to(noise))
show("inference path (fused kernels):", tot2, parts2)
raise SystemExit
print(os.environ.get("TAG", ""), f"total {float(tot):.3f} (ref fp32 {float(d['disc_total']):.3f}, ref bf16 {float(d['disc_total_bf16']):.3f});",
      f"r1 {float(parts['disc/r1_penalty']):.5f} (ref {float(d['disc_r1_penalty']):.5f}); r2 {float(parts['disc/r2_penalty']):.5f} (ref {float(d['disc_r2_penalty']):.5f})")
