"""Summarise a rocprofv3 (ROCm 7) rocpd SQLite result: per-kernel calls / total / mean / min / max.

    rocprofv3 --kernel-trace --stats -d out -o run -- python3 script.py ...
    python scripts/rocpd_kernel_stats.py out/run_results.db [--csv summary.csv] [--skip-first N]
"""
import argparse
import csv
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--skip-first", type=int, default=0, help="ignore the first N dispatches of every kernel (warm-up)")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall() \
        if _has_cols(db) else _fallback(db)
    per = {}
    for name, start, end, grid, wg in rows:
        per.setdefault(name, []).append((end - start, grid, wg))
    out = []
    for name, v in per.items():
        v = v[args.skip_first:] or v
        d = [x[0] for x in v]
        out.append((name, len(d), sum(d) / 1e3, sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, v[0][1], v[0][2]))
    out.sort(key=lambda r: -r[2])
    hdr = ("kernel", "calls", "total_us", "mean_us", "min_us", "max_us", "grid_x", "workgroup_x")
    if args.csv:
        with open(args.csv, "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(hdr)
            for r in out:
                wr.writerow([r[0], r[1]] + [f"{x:.2f}" for x in r[2:6]] + list(r[6:]))
    print(f"{'calls':>6} {'total_us':>11} {'mean_us':>10} {'min_us':>10} {'max_us':>10} {'grid':>8} {'wg':>5}  kernel")
    for r in out:
        print(f"{r[1]:6d} {r[2]:11.1f} {r[3]:10.1f} {r[4]:10.1f} {r[5]:10.1f} {r[6]:8d} {r[7]:5d}  {r[0][:110]}")


def _has_cols(db):
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    return {"name", "start", "end", "grid_x", "workgroup_x"} <= set(cols)


def _fallback(db):
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    sys.stderr.write(f"kernels view columns: {cols}\n")
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gx = [c for c in cols if c.startswith("grid") and c.endswith("x")][0]
    wx = [c for c in cols if c.startswith("workgroup") and c.endswith("x")][0]
    return db.execute(f"select {name}, start, end, {gx}, {wx} from kernels order by start").fetchall()


if __name__ == "__main__":
    main()
