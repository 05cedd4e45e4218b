#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc SQ_* counter CSVs (one or more passes) -> CSV on stdout / file.

    python tools/sq_summarize.py <counter_collection.csv> [...] [--match k_attn] [--skip-first N] [--out profiles/x.csv]

Units (MI355X_MICROARCH.md, cycle-constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count
quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs.  mfma_busy_frac below =
SQ_VALU_MFMA_BUSY_CYCLES / (duration x clock x 1024 SIMDs) with the clock taken from GRBM_GUI_ACTIVE when collected.
"""
import argparse
import collections
import csv
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--match", default="")
    ap.add_argument("--skip-first", type=int, default=8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for path in a.files:
        seen = collections.Counter()
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if a.match and a.match not in name:
                    continue
                key = (name, row["Counter_Name"])
                seen[key] += 1
                if seen[key] <= a.skip_first:
                    continue
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                acc[name]["_dur_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
                meta[name] = (row["Grid_Size"], row["Workgroup_Size"], row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"])
    out = open(a.out, "w", newline="") if a.out else sys.stdout
    w = csv.writer(out)
    counters = sorted({c for v in acc.values() for c in v if not c.startswith("_")})
    w.writerow(["kernel", "grid", "wg", "vgpr", "agpr", "sgpr", "lds", "scratch", "dispatches", "avg_dur_us"] + counters + ["mfma_busy_frac_at_2.4GHz", "clock_GHz_from_GRBM"])
    for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]["_dur_ns"])):
        dur = sum(v["_dur_ns"]) / len(v["_dur_ns"])
        avg = {c: sum(v[c]) / len(v[c]) for c in counters if v.get(c)}
        n = max(len(v[c]) for c in counters if v.get(c))
        mf = avg.get("SQ_VALU_MFMA_BUSY_CYCLES")
        frac = mf / (dur * 2.4 * 1024) if mf else ""
        clk = avg.get("GRBM_GUI_ACTIVE", 0) / 8 / dur if avg.get("GRBM_GUI_ACTIVE") else ""
        w.writerow([name[:100], *meta[name], n, f"{dur / 1e3:.2f}"] + [f"{avg.get(c, 0):.0f}" for c in counters] + [f"{frac:.4f}" if frac != "" else "", f"{clk:.3f}" if clk != "" else ""])


if __name__ == "__main__":
    main()
