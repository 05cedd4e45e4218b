"""Diagnostic (GPU box): where does k_attn_swp's output go non-finite / wrong for a spiked key?  python tests/probes/swp_probe.py spike key [key ...]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import titok_oracle as O
from titok_video_amd import _lib
from titok_video_amd.plan import BatchPlan
DEV = torch.device("cuda:0")
spike = float(sys.argv[1]); keys = [int(a) for a in sys.argv[2:]]
plan = BatchPlan([(16, 64, 64)], [9], (4, 8, 8), DEV)
hq, hkv, d, gq = 4, 2, 256, 128
ld = 2 * d + 2 * gq
g = torch.Generator().manual_seed(3)
x = torch.randn(plan.total_rows, ld, generator=g) * 0.5
q = x[:, :d].view(-1, 4, 64)
for n, key in enumerate(keys):
    x[key, 2 * d: 2 * d + gq] = (spike + 8.0 * n) * torch.sign(q[5, 0]).repeat(2)
x[140, :64] = q[5, 0]
q_f32 = x[:, :d].clone()
x = x.to(torch.bfloat16)
xd = x.to(DEV)
xd[:, :d] = (q_f32 * (0.125 * 1.4426950408889634)).to(torch.bfloat16).to(DEV)
tab = plan.attention_table(hq, hkv, False)
f = x.float()
qq, gt, k, v = f.split([d, d, gq, gq], dim=-1)
ref = O.attention_varlen(qq.unflatten(-1, (hq, 64)), k.unflatten(-1, (hkv, 64)), v.unflatten(-1, (hkv, 64)), plan.cu_seqlens).flatten(-2)
sc = (xd[:, :d].float().cpu().view(-1, 4, 64)[:, :, :] )
for flags, name in ((4 | 8, "swp"), (4, "bf16")):
    out = torch.full((plan.total_rows, d), float("nan"), dtype=torch.bfloat16, device=DEV)
    _lib.check(_lib.lib().ttv_attention(xd.data_ptr(), ld, out.data_ptr(), d, plan.cu_dev.data_ptr(), tab.data_ptr(), tab.shape[0], hq, hkv, 64, flags,
                                        _lib.TTV_BF16, _lib.stream_ptr(DEV)), "attention")
    torch.cuda.synchronize()
    o = out.float().cpu()
    bad = ~torch.isfinite(o)
    rows = bad.any(1).nonzero().flatten().tolist()
    print(name, "non-finite rows:", len(rows), rows[:40])
    for r in rows[:6]:
        heads = [h for h in range(4) if bad[r, h * 64:(h + 1) * 64].any()]
        for h in heads:
            s = (sc[r, h] @ k.view(-1, 2, 64)[:, h // 2].T)          # exponents (log2 units)
            print(f"  row {r} head {h}: bad cols {int(bad[r, h*64:(h+1)*64].sum())}; score max first tile {float(s[:64].max()):.1f}; per tile max",
                  [round(float(s[i:i + 64].max()), 1) for i in range(0, s.numel(), 64)], "values", o[r, h * 64:h * 64 + 4].tolist())
    fin = torch.isfinite(o)
    err = (o - ref).abs()
    err[~fin] = 0
    print(name, "max abs err over finite:", float(err.max()))
