#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs into per-kernel averages and profiles/pmc_traffic.json.

    python tools/pmc_summarize.py --fetch <..._counter_collection.csv> --write <..._counter_collection.csv> \
        [--skip-first N] --tag r01_final --out profiles

FETCH_SIZE / WRITE_SIZE are collected in SEPARATE passes and are in KB; on gfx950 FETCH_SIZE reports half of wide
coalesced reads, so it is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section).  The per-launch traffic of a kernel
class is fetch*2 + write, averaged over its steady-state dispatches (the first --skip-first dispatches per kernel are
warm-up: cold L2 / first-touch).
"""
import argparse
import collections
import csv
import json
import os

CLASS_OF = (            # kernel-name fragment -> bench.py --kernel class
    ("k_attn_swp<", "attention"),     # round 5: the software-pipelined kernel (tables of full items, pre-scaled q)
    ("k_attn_bf16<", "attention"),
    ("k_qkv256<", "gemm_qkv"),        # round 4: the wave-pipelined to_qkv kernel (k_gemm_k256<1,..> before / with TTV_QKV256=0)
    ("k_gemm_k256<1,", "gemm_qkv"),
    ("k_gemm_k256<2,", "gemm_geglu"),
    ("k_mlp256<", "layer_tail"),
    ("k_gemm_rowtile_norm", "gemm_w3_keel"),
    ("k_gemm_k256_rownorm", "gemm_out_keel"),
)


def per_kernel(path, counter, skip_first):
    seen = collections.Counter()
    acc = collections.defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            seen[name] += 1
            if seen[name] > skip_first:
                acc[name].append(float(row["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items() if v}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--skip-first", type=int, default=16)
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--out", default="profiles")
    ap.add_argument("--note", default="")
    args = ap.parse_args()

    fetch = per_kernel(args.fetch, "FETCH_SIZE", args.skip_first)
    write = per_kernel(args.write, "WRITE_SIZE", args.skip_first)
    os.makedirs(args.out, exist_ok=True)
    for nm, tab, col in (("fetch", fetch, "avg_FETCH_SIZE_KB_raw"), ("write", write, "avg_WRITE_SIZE_KB")):
        with open(os.path.join(args.out, f"{args.tag}_pmc_{nm}_per_kernel.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "dispatches", col])
            for k, (n, v) in sorted(tab.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
                w.writerow([k[:120], n, f"{v:.1f}"])

    import subprocess
    try:     # the tree the counters were collected on (bench.py prints it as roofline.traffic_source)
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or "unknown"
    except Exception:
        commit = "unknown"
    traffic = {"commit": commit, "tag": args.tag, "_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units); "
                        "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); "
                        f"steady-state dispatches only (first {args.skip_first} per kernel dropped); source "
                        f"profiles/{args.tag}_pmc_*_per_kernel.csv. {args.note}".strip()}
    for cls in dict.fromkeys(c for _, c in CLASS_OF):          # a class may have several kernel names: one weighted average over all of them
        prefixes = [pf for pf, c in CLASS_OF if c == cls]
        fk = [(n, v) for k, (n, v) in fetch.items() if any(pf in k for pf in prefixes)]
        wk = [(n, v) for k, (n, v) in write.items() if any(pf in k for pf in prefixes)]
        if not fk or not wk:
            continue
        fb = 2.0 * 1024.0 * sum(n * v for n, v in fk) / sum(n for n, _ in fk)
        wb = 1024.0 * sum(n * v for n, v in wk) / sum(n for n, _ in wk)
        traffic[cls] = fb + wb
        traffic[cls + "_detail"] = {"fetch_bytes_x2_corrected": fb, "write_bytes": wb, "launches": sum(n for n, _ in fk)}
    with open(os.path.join(args.out, "pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps({k: v for k, v in traffic.items() if not k.startswith("_") and not k.endswith("_detail")}))


if __name__ == "__main__":
    main()
