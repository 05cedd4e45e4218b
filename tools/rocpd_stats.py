#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd SQLite result (the default output format when --output-format is not csv).

    python tools/rocpd_stats.py <results.db> [--per N] [--csv out.csv]

--per N divides call counts / totals by N (e.g. the number of profiled steps) for a per-step view.
"""
import argparse
import csv
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--per", type=float, default=1.0)
    ap.add_argument("--csv")
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = list(db.execute(
        f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"))
    total = sum(r[2] for r in rows)
    out = [("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for name, n, tot, mn, mx in rows:
        out.append((name[:110], f"{n / args.per:.1f}", f"{tot / 1e3 / args.per:.1f}", f"{tot / n / 1e3:.1f}",
                    f"{mn / 1e3:.1f}", f"{mx / 1e3:.1f}", f"{100.0 * tot / total:.2f}"))
    if args.csv:
        with open(args.csv, "w", newline="") as f:
            csv.writer(f).writerows(out)
    for r in out[:args.top + 1]:
        print(f"{r[2]:>10} {r[1]:>7} {r[3]:>8} {r[6]:>6}  {r[0]}")
    print(f"total {total / 1e3 / args.per:.1f} us")


if __name__ == "__main__":
    main()
