#!/usr/bin/env python3
"""Index-exact inference alone (profiling target): TiTok.forward after set_index_exact(MODE) on the benchmark batch.
    MODE=split3|fp32 STEPS=20 python tools/exact_index_bench.py        (rocprofv3 --kernel-trace --stats -- python3 tools/exact_index_bench.py)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

mode = os.environ.get("MODE", "split3")
steps = int(os.environ.get("STEPS", "20"))
wl = bench.WORKLOADS["tiny"]
dev = torch.device("cuda", 0)
sd = bench.seeded_titok_state(0, wl["size"], wl["size"], gain=wl["gain"])
if os.environ.get("DEBUG_BITS"):      # e.g. 256 = 128-token GEMM tiles everywhere, 128 = 160-token tiles (ttv_debug_set, this thread)
    from titok_video_amd import _lib
    _lib.lib().ttv_debug_set(int(os.environ["DEBUG_BITS"]))
leg = bench.exact_index_leg(wl, sd, dev, 0, mode, steps=steps)
print({k: v for k, v in leg.items() if k not in ("dtype", "what", "reference")})
