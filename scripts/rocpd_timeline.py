"""Timeline of the last N kernel dispatches of a rocprofv3 rocpd SQLite result: start offset, duration and the idle
gap in front of each kernel (microseconds), plus per-kernel totals over that window.

    python scripts/rocpd_timeline.py out/run_results.db --last 400 [--full]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--last", type=int, default=200)
    ap.add_argument("--full", action="store_true", help="print every dispatch, not only the totals")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    try:
        rows = db.execute("select name, start, end from kernels order by start").fetchall()
    except sqlite3.OperationalError:
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
        rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    rows = rows[-args.last:]
    t0 = rows[0][1]
    per, gaps, prev_end = {}, 0.0, None
    for name, st, en in rows:
        gap = 0.0 if prev_end is None else max(0.0, (st - prev_end) / 1e3)
        gaps += gap
        short = name.split("(")[0][-70:]
        if args.full:
            print(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:8.1f} gap {gap:6.1f}  {short}")
        d = per.setdefault(short, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += (en - st) / 1e3
        d[2] += gap
        prev_end = en if prev_end is None else max(prev_end, en)
    span = (max(r[2] for r in rows) - t0) / 1e3
    print(f"window: {len(rows)} dispatches, span {span:.1f} us, idle gaps {gaps:.1f} us")
    print(f"{'calls':>6} {'busy_us':>10} {'mean_us':>9} {'gap_before_us':>14} {'mean_gap':>9}  kernel")
    for k, (n, busy, gap) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:6d} {busy:10.1f} {busy / n:9.2f} {gap:14.1f} {gap / n:9.2f}  {k}")


if __name__ == "__main__":
    main()
