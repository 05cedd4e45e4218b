#!/usr/bin/env python3
"""CPU experiment (test infrastructure; imports oracle/): what accuracy does a split-bf16 ENCODER keep?

Every matrix product of the oracle encoder (linears, q k^T, p v) is replaced by an emulation of the n-pass split scheme
    x = x0 + x1 (+ x2),  x_i = bf16(x - sum_{j<i} x_j);   a b ~ sum over the kept (i, j) pairs of a_i b_j   (fp32 accumulation)
and the pre-rounding FSQ values are compared with the reference's fp32 run (tests/golden/titok_cfg1.npz).  Passes:
    1 = (0,0)                    plain bf16 operands
    3 = (0,0) (0,1) (1,0)        "bf16x3"
    4 = + (1,1)
    6 = three terms: (0,0) (0,1) (1,0) (0,2) (2,0) (1,1)   "bf16x6"
Everything that is not a matrix product (norms, rotary, softmax, GELU, residual stream) stays fp32, as in the fp32 towers.
    python tests/probes/split_bf16_probe.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import titok_oracle as O  # noqa: E402
from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips  # noqa: E402

PAIRS = {1: [(0, 0)], 3: [(0, 0), (0, 1), (1, 0)], 4: [(0, 0), (0, 1), (1, 0), (1, 1)], 6: [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]}


def split(x, n):
    parts, rest = [], x.float()
    for _ in range(n):
        p = rest.to(torch.bfloat16).float()
        parts.append(p)
        rest = rest - p
    return parts


def mm(a, b, passes):
    """a [.., M, K] @ b [.., K, N] with split operands; each partial product accumulates in fp32 (torch CPU fp32 matmul)."""
    if passes == 0:
        return a @ b
    n = 3 if passes == 6 else 2 if passes > 1 else 1
    pa, pb = split(a, n), split(b, n)
    out = None
    for i, j in reversed(PAIRS[passes]):           # small terms first
        t = pa[i] @ pb[j]
        out = t if out is None else out + t
    return out


def run(passes, where):
    lin0, attn0 = F.linear, O.attention_varlen

    def linear(x, w, b=None):
        y = mm(x.float(), w.float().t(), passes if "linear" in where else 0)
        return y if b is None else y + b.float()

    def attention(q, k, v, cu):
        L, Hq, D = q.shape
        rep = Hq // k.shape[1]
        out = torch.empty_like(q, dtype=torch.float32)
        p_att = passes if "attn" in where else 0
        for bi in range(len(cu) - 1):
            s, e = int(cu[bi]), int(cu[bi + 1])
            qb = q[s:e].float().transpose(0, 1)
            kb = k[s:e].float().transpose(0, 1).repeat_interleave(rep, dim=0)
            vb = v[s:e].float().transpose(0, 1).repeat_interleave(rep, dim=0)
            sc = mm(qb, kb.transpose(1, 2), p_att) * D ** -0.5
            pt = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
            out[s:e] = (mm(pt, vb, p_att) / pt.sum(dim=-1, keepdim=True)).transpose(0, 1)
        return out
    F.linear, O.attention_varlen = linear, attention
    try:
        fix = np.load(os.path.join(ROOT, "tests", "golden", "titok_cfg1.npz"))
        sd = seeded_titok_state(0, gain=6.0)
        clips = synthetic_clips([(16, 128, 128)] * 4, seed=1234)
        with torch.no_grad():
            z = O.encoder_forward(clips, [128] * 4, sd, "tiny", prefix="encoder.")
            b = O.fsq_bound(z.float(), [7, 5, 5, 5, 5])
    finally:
        F.linear, O.attention_varlen = lin0, attn0
    ref_b = torch.from_numpy(fix["bounded"])
    ref_i = torch.from_numpy(fix["indices"])
    idx = O.fsq_forward(z.float(), [7, 5, 5, 5, 5])[1]
    margin = O.fsq_margin(ref_b)
    err = (b - ref_b).abs()
    out = {"passes": passes, "where": where, "mean_err": float(err.mean()), "max_err": float(err.max()), "index_match": float((idx == ref_i).float().mean())}
    for tau in (1e-3, 1e-2):
        safe = margin > tau
        out[f"mism_margin_gt_{tau:g}"] = f"{int((idx[safe] != ref_i[safe]).sum())}/{int(safe.sum())}"
    return out


if __name__ == "__main__":
    torch.set_num_threads(8)
    for passes, where in ((0, "linear+attn"), (1, "linear+attn"), (3, "linear+attn"), (4, "linear+attn"), (6, "linear+attn"), (3, "linear"), (3, "attn")):
        print(run(passes, where), flush=True)
