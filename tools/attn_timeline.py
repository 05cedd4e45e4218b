#!/usr/bin/env python3
"""When and where every block of the bf16 attention kernel ran, and at what shader clock (diagnostic build tools/attn_timeline.sh,
loaded through TTV_LIB_PATH): per table entry the 100 MHz constant clock and the shader clock at entry start / loop start / loop end
/ entry end, and the CU the block ran on.

    bash tools/attn_timeline.sh && TTV_LIB_PATH=titok_video_amd/csrc/build/libtitok_hip_timeline.so python3 tools/attn_timeline.py
    B=64 ... (clips)   TTV_ATTN_PERS=1 ... (persistent walk)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd import _lib  # noqa: E402
from titok_video_amd.plan import BatchPlan  # noqa: E402

DEV = torch.device("cuda:0")
lib = _lib.lib()
ST = _lib.stream_ptr(DEV)
B = int(os.environ.get("B", "32"))
plan = BatchPlan([(16, 128, 128)] * B, [128] * B, (4, 8, 8), DEV)
L = plan.total_rows
table = plan.attention_table(4, 2)
n = int(table.shape[0])
qkv = (torch.randn(L, 768, device=DEV) * 0.9).to(torch.bfloat16)
qkv[:, :256] = (qkv[:, :256].float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
out = torch.empty(L, 256, dtype=torch.bfloat16, device=DEV)
stamps = torch.zeros(n * 8, dtype=torch.int64, device=DEV)
lib.ttv_debug_stamps(stamps.data_ptr())
for _ in range(4):      # the last launch's stamps are the ones read (caches warm, clocks settled)
    _lib.check(lib.ttv_attention(qkv.data_ptr(), 768, out.data_ptr(), 256, plan.cu_dev.data_ptr(), table.data_ptr(), n, 4, 2, 64, 1 | 4, 0, ST), "attn")
torch.cuda.synchronize()
lib.ttv_debug_stamps(None)
s = stamps.view(n, 8).cpu()
live = s[:, 0] > 0
s = s[live]
real = s[:, :4].double()
t0 = real[:, 0].min()
us = (real - t0) / 100.0                       # 100 MHz -> microseconds since the first block started
core = (s[:, 5] - s[:, 4]).double()
dur = us[:, 3] - us[:, 0]
clk = core / (dur * 1e3)                       # shader cycles per nanosecond = GHz
hw, xcc = s[:, 6], s[:, 7] & 0xF
cu_key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
print(f"{B} clips, {int(live.sum())} table entries with a block; launch spans {float(us[:, 3].max()):.1f} us from the first block's start to the last block's end")
print(f"shader clock while the blocks ran (s_memtime / s_memrealtime): mean {float(clk.mean()):.3f} GHz, min {float(clk.min()):.3f}, max {float(clk.max()):.3f}")
print(f"per entry: prologue {float((us[:, 1] - us[:, 0]).mean()):.2f} us, key loop {float((us[:, 2] - us[:, 1]).mean()):.2f} us, epilogue + store acknowledgement {float((us[:, 3] - us[:, 2]).mean()):.2f} us, "
      f"total {float(dur.mean()):.2f} us (min {float(dur.min()):.2f}, max {float(dur.max()):.2f})")
modes = table.cpu()[:, 3][live]
for md, nm in ((0, "full items (128 rows)"), (1, "half items (64 rows, key range split between wave pairs)")):
    sel = modes == md
    if int(sel.sum()):
        print(f"  {nm}: {int(sel.sum())} entries, key loop {float((us[sel, 2] - us[sel, 1]).mean()):.2f} us, total {float(dur[sel].mean()):.2f} us")
print("per XCD: entries | mean shader clock GHz | mean entry time us | first-round entries' mean time | last end us")
for x in sorted(set(xcc.tolist())):
    sel = xcc == x
    first = sel & (us[:, 0] < 2.0)
    print(f"  XCD {int(x)}: {int(sel.sum()):5d} | {float(clk[sel].mean()):.3f} | {float(dur[sel].mean()):6.2f} | {float(dur[first].mean()) if int(first.sum()) else float('nan'):6.2f} | {float(us[sel, 3].max()):6.1f}")
edges = [0, 2, 5, 10, 15, 20, 25, 30, 35, 40, 45, 50, 55, 60, 70, 80, 100, 150, 1e9]
st_h = torch.histogram(us[:, 0].float(), torch.tensor(edges, dtype=torch.float32)).hist
en_h = torch.histogram(us[:, 3].float(), torch.tensor(edges, dtype=torch.float32)).hist
print("blocks starting / ending per interval (us since the first start):")
for i in range(len(edges) - 1):
    if st_h[i] or en_h[i]:
        print(f"  [{edges[i]:5.0f}, {edges[i + 1]:5.0f})   start {int(st_h[i]):5d}   end {int(en_h[i]):5d}")
cus, counts = torch.unique(cu_key, return_counts=True)
xs, xcounts = torch.unique(xcc, return_counts=True)
print(f"{len(cus)} distinct CUs ran blocks; entries per CU: min {int(counts.min())}, mean {float(counts.float().mean()):.2f}, max {int(counts.max())}; "
      f"entries per XCD: {[int(c) for c in xcounts]}")
# concurrency on a CU: for every block, how many blocks of the same CU overlap its loop midpoint
mid = (us[:, 1] + us[:, 2]) / 2
conc = torch.zeros(len(s))
for k in cus:
    idx = (cu_key == k).nonzero().flatten()
    for i in idx:
        conc[i] = float(((us[idx, 0] <= mid[i]) & (us[idx, 3] >= mid[i])).sum())
for c in sorted(set(conc.tolist())):
    sel = conc == c
    print(f"  blocks with {int(c)} resident on their CU at their loop midpoint: {int(sel.sum()):5d}   key loop {float((us[sel, 2] - us[sel, 1]).mean()):.2f} us")
# is the table's list-per-XCD assumption right?  entry i is meant for XCD i % 8
tix = live.nonzero().flatten()
same = ((tix % 8) == (tix[0] % 8)) 
x_of_list0 = xcc[same]
print(f"entries of list {int(tix[0] % 8)} ran on XCDs {sorted(set(x_of_list0.tolist()))}")
