#!/usr/bin/env python3
"""to_qkv + rotary at the benchmark shape (36 864 tokens, K = 256, N = 768): the weight-stationary kernel k_qkv256ws (ttv_qkv256ws.inc) and the streaming
wave-pipelined kernel k_qkv256 (ttv_qkv256.inc, the default) against k_gemm_k256<EPI_QKV_ROPE> (ttv_debug_set bit 15; bit 17 selects the weight-stationary kernel), through the C-ABI entry `ttv_linear_qkv_rope`
(table path, no folded pre-norm).  Prints microseconds per launch for both, A/B interleaved, and the largest difference of the two
outputs (both round the same fp32 products to bf16; rstd = 1 on this path, so they must agree to bf16 rounding of the rotary sums).
B= clips (default 32), also a ragged shape whose last token tile is partial.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from titok_video_amd import _lib  # noqa: E402
from titok_video_amd.plan import BatchPlan  # noqa: E402

DEV = torch.device("cuda:0")
lib = _lib.lib()
S = _lib.stream_ptr(DEV)
OLD, STREAM, WS = 32768, 0, 131072


def t(fn, it=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


def run(shapes, counts, reps):
    plan = BatchPlan(shapes, counts, (4, 8, 8), DEV)
    L, d, g = plan.total_rows, 256, 128
    bf = torch.bfloat16
    torch.manual_seed(0)
    x = torch.randn(L, d, device=DEV).to(bf)
    w = (torch.randn(2 * d + 2 * g, d, device=DEV) * d ** -0.5).to(bf)
    outs = {}
    for name, dbg in (("k_gemm_k256", OLD), ("k_qkv256", STREAM), ("k_qkv256ws", WS)):
        qkv = torch.full((L, 2 * d + 2 * g), float("nan"), dtype=bf, device=DEV)
        lib.ttv_debug_set(dbg)
        _lib.check(lib.ttv_linear_qkv_rope(x.data_ptr(), d, w.data_ptr(), d, qkv.data_ptr(), 2 * d + 2 * g, L, d, g, plan.rope_cs.data_ptr(), 0, S), "qkv")
        torch.cuda.synchronize()
        outs[name] = qkv.float()
    lib.ttv_debug_set(0)
    for new in ("k_qkv256", "k_qkv256ws"):
        diff = (outs[new] - outs["k_gemm_k256"]).abs()
        print(f"rows {L}: {new}: max |new - old| {float(torch.nan_to_num(diff, nan=1e30).max()):.3e}, differing elements {int((diff != 0).sum())} of {diff.numel()}, "
              f"nan in new {int(torch.isnan(outs[new]).sum())}", flush=True)
    qkv = torch.empty(L, 2 * d + 2 * g, dtype=bf, device=DEV)

    def call():
        _lib.check(lib.ttv_linear_qkv_rope(x.data_ptr(), d, w.data_ptr(), d, qkv.data_ptr(), 2 * d + 2 * g, L, d, g, plan.rope_cs.data_ptr(), 0, S), "qkv")

    for _ in range(reps):
        for name, dbg in (("k_gemm_k256", OLD), ("k_qkv256", STREAM), ("k_qkv256 no stores", 1), ("k_qkv256ws", WS), ("k_qkv256ws no stores", WS | 1)):
            if os.environ.get("ONLY_WS") and not name.startswith("k_qkv256ws"):
                continue
            lib.ttv_debug_set(dbg)
            print(f"  {name:22s} {t(call):7.1f} us", flush=True)
    lib.ttv_debug_set(0)


B = int(os.environ.get("B", "32"))
run([(8, 32, 48), (4, 16, 24), (4, 24, 40)], [3, 5, 7], 0)      # ragged: partial last tile
run([(16, 128, 128)] * B, [128] * B, 3)
