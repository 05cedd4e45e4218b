"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own code.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py

* model/quantizer/fsq.py, model/base/rope.py, train_utils/codebook_logging.py are imported unmodified.
* model/titok.py (+ blocks/transformer/utils) hard-import two third-party packages that are absent
  here (flash_attn, xformers - unpinned in the reference tree).  They are replaced, before import, by
  local stand-in modules stating the published definitions:
    RMSNorm(hidden, eps=1e-5): fp32 x*rsqrt(mean(x^2)+eps)*w, no bias, output in input dtype
    flash_attn_varlen_func: per-sequence non-causal softmax(q k^T D^-0.5) v in fp32, GQA, output in q dtype
    SwiGLU: inert placeholder (never instantiated by the reference).
  The reference's own modules then run unmodified on CPU (fp32 under no_grad).
* Weights come from the seed recipe in titok_video_amd/synthetic.py (loaded via load_state_dict), so
  the fixtures hold inputs/outputs only.  No reference source is copied into this repository.
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)


def install_standins():
    import torch.nn as nn

    class RMSNorm(nn.Module):
        def __init__(self, hidden_size, eps=1e-5, **kw):
            super().__init__()
            self.eps = eps
            self.weight = nn.Parameter(torch.ones(hidden_size))
            self.bias = None

        def forward(self, x):
            xf = x.float()
            y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps) * self.weight.float()
            return y.to(x.dtype)

    def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, **kw):
        hq, hkv, d = q.shape[1], k.shape[1], q.shape[2]
        out = torch.empty_like(q)
        cu = cu_seqlens_q.tolist()
        for b in range(len(cu) - 1):
            s, e = cu[b], cu[b + 1]
            qb = q[s:e].float().transpose(0, 1)
            kb = k[s:e].float().transpose(0, 1).repeat_interleave(hq // hkv, 0)
            vb = v[s:e].float().transpose(0, 1).repeat_interleave(hq // hkv, 0)
            p = torch.softmax(qb @ kb.transpose(1, 2) * d ** -0.5, dim=-1)
            out[s:e] = (p @ vb).transpose(0, 1).to(q.dtype)
        return out

    fa = types.ModuleType("flash_attn")
    fa.flash_attn_varlen_func = flash_attn_varlen_func
    ops = types.ModuleType("flash_attn.ops")
    tri = types.ModuleType("flash_attn.ops.triton")
    ln = types.ModuleType("flash_attn.ops.triton.layer_norm")
    ln.RMSNorm = RMSNorm
    xf = types.ModuleType("xformers")
    xops = types.ModuleType("xformers.ops")
    xops.SwiGLU = type("SwiGLU", (nn.Module,), {})
    for name, mod in [("flash_attn", fa), ("flash_attn.ops", ops), ("flash_attn.ops.triton", tri),
                      ("flash_attn.ops.triton.layer_norm", ln), ("xformers", xf), ("xformers.ops", xops)]:
        sys.modules[name] = mod


def ref_config(levels, enc="tiny", dec="tiny", patch=(4, 8, 8)):
    return SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(
        patch_size=list(patch), fsq_levels=list(levels), encoder_size=enc, decoder_size=dec)))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def np32(t):
    return t.detach().to(torch.float32).cpu().numpy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from titok_video_amd.synthetic import seeded_titok_state, synthetic_clips

    # ---------------- FSQ known-answer tests (reference fsq.py, unmodified) ----------------
    from model.quantizer.fsq import FSQ
    fsq_out = {}
    for tag, levels in [("a", [7, 5, 5, 5, 5]), ("b", [8, 8, 8, 6, 5])]:
        f = FSQ(levels)
        # dense sweep crossing every rounding boundary of every channel + random vectors
        sweep = torch.linspace(-4.0, 4.0, 4001)
        z1 = torch.stack([sweep.roll(17 * c) for c in range(len(levels))], dim=-1)
        g = torch.Generator().manual_seed(11)
        z2 = torch.randn(4096, len(levels), generator=g) * 1.5
        z = torch.cat([z1, z2], 0)
        codes, d = f(z)
        fsq_out[f"levels_{tag}"] = np.array(levels, dtype=np.int32)
        fsq_out[f"z_{tag}"] = np32(z)
        fsq_out[f"codes_{tag}"] = np32(codes)
        fsq_out[f"indices_{tag}"] = d["indices"].numpy()
        fsq_out[f"bounded_{tag}"] = np32(f.bound(z))
        all_idx = torch.arange(f.codebook_size, dtype=torch.int32)
        fsq_out[f"codebook_{tag}"] = np32(f.indices_to_codes(all_idx))
        assert torch.equal(f.codes_to_indices(f.indices_to_codes(all_idx)), all_idx)
    save("fsq_kat.npz", **fsq_out)

    # ---------------- RoPE table + rotary apply (reference rope.py, unmodified) ----------------
    from model.base.rope import RoPE, apply_rotary_emb
    rope = RoPE(head_dim=64, grid_dims=3)
    rope_out = {}
    cases = [([[1, 2, 2]], [1]), ([[2, 4, 6], [1, 2, 3]], [3, 5]), ([[4, 16, 16]], [128]), ([[2, 3, 5], [1, 1, 1], [3, 2, 2]], [7, 1, 4])]
    for i, (grids, counts) in enumerate(cases):
        fc = rope(torch.tensor(grids, dtype=torch.int32), torch.tensor(counts, dtype=torch.int32), torch.device("cpu"))
        rope_out[f"grids_{i}"] = np.array(grids, dtype=np.int32)
        rope_out[f"counts_{i}"] = np.array(counts, dtype=np.int32)
        rope_out[f"cos_{i}"] = fc.real.numpy().astype(np.float64)
        rope_out[f"sin_{i}"] = fc.imag.numpy().astype(np.float64)
    g = torch.Generator().manual_seed(5)
    grids, counts = cases[1]
    fc = rope(torch.tensor(grids, dtype=torch.int32), torch.tensor(counts, dtype=torch.int32), torch.device("cpu"))
    q = torch.randn(fc.shape[0], 4, 64, generator=g)
    with torch.no_grad():
        rope_out["rot_q"] = np32(q)
        rope_out["rot_out"] = np32(apply_rotary_emb(q.clone(), fc))
    save("rope_kat.npz", **rope_out)

    # ---------------- CodebookLogger (reference codebook_logging.py, unmodified) ----------------
    from train_utils.codebook_logging import CodebookLogger
    g = torch.Generator().manual_seed(9)
    cb = CodebookLogger(64)
    samples = [torch.randint(0, 64, (int(n),), generator=g, dtype=torch.int32)
               for n in torch.randint(1, 9, (80,), generator=g)]
    cb(samples)     # FIFO keeps the last 64
    sc = cb.get_scores()
    save("codebook_kat.npz", sizes=np.array([len(s) for s in samples], dtype=np.int32),
         flat=torch.cat(samples).numpy(), codebook_size=np.int32(64),
         usage=np.float64(float(sc["codebook/usage_percent"])), entropy=np.float64(float(sc["codebook/entropy"])))

    # ---------------- towers / TiTok (reference modules + third-party stand-ins) ----------------
    install_standins()
    from model.titok import TiTok
    from model.base.utils import patch_rearrange, unpatch_rearrange

    # patch / unpatch on an index-valued tensor
    clip = torch.arange(3 * 8 * 16 * 24, dtype=torch.float32).reshape(3, 8, 16, 24)
    p = patch_rearrange((4, 8, 8))(clip)
    back = unpatch_rearrange((4, 8, 8))(p, torch.tensor([2, 2, 3]))
    assert torch.equal(back, clip)
    save("patch_kat.npz", clip=np32(clip), patches=np32(p))

    levels = [7, 5, 5, 5, 5]
    model = TiTok(ref_config(levels)).eval()
    sd = seeded_titok_state(seed=0)
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    print("load_state_dict:", missing)

    def run(shapes, counts, seed):
        clips = synthetic_clips(shapes, seed=seed)
        tc = torch.tensor(counts, dtype=torch.int32)
        with torch.no_grad():
            grids = torch.tensor([c.shape[1:] for c in clips], dtype=torch.int32)
            z = model.encoder(clips, tc, grids)
            codes, d = model.quantize(z)
            bounded = model.quantize.bound(z.float())
            recon = model.decode(codes, tc, grids)
            recon2 = model.decode_indices(d["indices"], grids, tc)
        for a, b in zip(recon, recon2):
            assert torch.equal(a, b)
        margin = (0.5 - (bounded - bounded.round()).abs()).min(-1).values
        print(f"  shapes={shapes} K={counts}: distinct idx {len(set(d['indices'].tolist()))}/{len(d['indices'])}, "
              f"min margin {margin.min():.2e}, recon std {torch.cat([r.flatten() for r in recon]).std():.3f}")
        return clips, z, codes, d["indices"], bounded, recon

    model_bf16 = TiTok(ref_config(levels)).eval()
    model_bf16.load_state_dict(sd, strict=True)
    model_bf16 = model_bf16.to(torch.bfloat16)

    def run_bf16(shapes, counts, seed, ref_codes):
        """The reference's own modules in bf16 (its real configuration is bf16 on GPU): how far ITS bf16 run is from
        its fp32 run on the same inputs - the yardstick for the HIP bf16 path."""
        clips = synthetic_clips(shapes, seed=seed, dtype=torch.bfloat16)
        tc = torch.tensor(counts, dtype=torch.int32)
        with torch.no_grad():
            grids = torch.tensor([c.shape[1:] for c in clips], dtype=torch.int32)
            z = model_bf16.encoder(clips, tc, grids)
            _, d = model_bf16.quantize(z)
            bounded = model_bf16.quantize.bound(z.float())
            recon = model_bf16.decode(ref_codes.to(torch.bfloat16), tc, grids)   # decoder on the fp32 run's codes
        return d["indices"], bounded, recon

    # small mixed-shape batch: full tensors
    shapes = [(4, 16, 16), (8, 32, 48), (4, 8, 24), (8, 16, 16)]
    counts = [1, 5, 3, 8]
    clips, z, codes, idx, bounded, recon = run(shapes, counts, seed=77)
    small = {"shapes": np.array(shapes, dtype=np.int32), "counts": np.array(counts, dtype=np.int32),
             "clip_seed": np.int32(77), "weight_seed": np.int32(0), "levels": np.array(levels, dtype=np.int32),
             "z": np32(z), "codes": np32(codes), "indices": idx.numpy(), "bounded": np32(bounded)}
    for i, r in enumerate(recon):
        small[f"recon_{i}"] = np32(r)
    i16, b16, r16 = run_bf16(shapes, counts, 77, codes)
    small["indices_refbf16"], small["bounded_refbf16"] = i16.numpy(), np32(b16)
    for i, r in enumerate(r16):
        small[f"recon_refbf16_{i}"] = np32(r)
    save("titok_small.npz", **small)

    # packing invariance material: clip 1 alone
    clips1, z1, codes1, idx1, bounded1, recon1 = run([shapes[1]], [counts[1]], seed=78)
    save("titok_single.npz", shape=np.array(shapes[1], dtype=np.int32), count=np.int32(counts[1]),
         clip_seed=np.int32(78), weight_seed=np.int32(0), z=np32(z1), indices=idx1.numpy(),
         bounded=np32(bounded1), recon=np32(recon1[0]))

    # BASELINE config #1: 4 clips 16x128x128, K=128 (recon stored as a strided sample)
    shapes = [(16, 128, 128)] * 4
    counts = [128] * 4
    clips, z, codes, idx, bounded, recon = run(shapes, counts, seed=1234)
    rs = torch.stack(recon)                       # [4,3,16,128,128]
    sample = rs[:, :, ::4, ::8, ::8].contiguous()   # [4,3,4,16,16]
    i16, b16, r16 = run_bf16(shapes, counts, 1234, codes)
    sample16 = torch.stack(r16)[:, :, ::4, ::8, ::8].contiguous()
    save("titok_cfg1.npz", indices_refbf16=i16.numpy(), bounded_refbf16=np32(b16), recon_sample_refbf16=np32(sample16), shapes=np.array(shapes, dtype=np.int32), counts=np.array(counts, dtype=np.int32),
         clip_seed=np.int32(1234), weight_seed=np.int32(0), levels=np.array(levels, dtype=np.int32),
         z=np32(z), indices=idx.numpy(), bounded=np32(bounded), recon_sample=np32(sample),
         recon_mean=np.float64(rs.double().mean().item()), recon_std=np.float64(rs.double().std().item()),
         recon_abs_sum=np.float64(rs.double().abs().sum().item()))

    # single sub-blocks on a small packed batch (Attn, GEGLU, one pre-LN layer + one KEEL layer)
    from model.base.transformer import Attn, GEGLU
    enc = model.encoder
    grids = torch.tensor([[2, 4, 6], [1, 2, 3]], dtype=torch.int32)
    tcs = torch.tensor([3, 5], dtype=torch.int32)
    seq = (grids.prod(-1) + tcs)
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), seq.cumsum(0).to(torch.int32)])
    fc = enc.rope(grids, tcs, torch.device("cpu"))
    g = torch.Generator().manual_seed(21)
    x = torch.randn(int(cu[-1]), 256, generator=g)
    with torch.no_grad():
        a1 = enc.model_layers.attn_layer[1](x, fc, cu, seq.max())
        f1 = enc.model_layers.ffd_layer[1](x)
        full = enc.model_layers(x, fc, cu, seq.max())
    save("blocks_kat.npz", grids=grids.numpy(), counts=tcs.numpy(), x=np32(x), attn1=np32(a1), ffd1=np32(f1),
         stack=np32(full), weight_seed=np.int32(0))


if __name__ == "__main__":
    main()
