"""Per-kernel means of the hardware counters in rocprofv3 (ROCm 7) rocpd SQLite results.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d out -o a -- python3 script.py
    python scripts/rocpd_pmc_summary.py out/a_results.db [more.db ...] [--match gemm]

Counters are summed over their hardware instances per dispatch, then averaged over the dispatches
of a kernel (the first --skip-first dispatches of each kernel are ignored)."""
import argparse
import collections
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dbs", nargs="+")
    ap.add_argument("--match", default="")
    ap.add_argument("--skip-first", type=int, default=2)
    args = ap.parse_args()
    table = collections.defaultdict(dict)       # kernel -> {counter: mean, "us": mean duration, "n": dispatches}
    for path in args.dbs:
        db = sqlite3.connect(path)
        per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, dispatch) -> counter -> sum
        dur = {}
        for name, disp, cname, val, d in db.execute(
                "select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
            if args.match and args.match not in name:
                continue
            per[(name, disp)][cname] += val
            dur[(name, disp)] = d
        by_kernel = collections.defaultdict(list)
        for (name, disp) in sorted(per, key=lambda k: k[1]):
            by_kernel[name].append((name, disp))
        for name, keys in by_kernel.items():
            keys = keys[args.skip_first:] or keys
            row = table[name]
            row["n"] = len(keys)
            row["us"] = sum(dur[k] for k in keys) / len(keys) / 1e3
            for c in per[keys[0]]:
                row[c] = sum(per[k][c] for k in keys) / len(keys)
    for name, row in sorted(table.items(), key=lambda kv: -kv[1].get("us", 0)):
        print(name[:100])
        print("   " + "  ".join(f"{k}={v:,.1f}" if isinstance(v, float) else f"{k}={v}" for k, v in row.items()))


if __name__ == "__main__":
    main()
