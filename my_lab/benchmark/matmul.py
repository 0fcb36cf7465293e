"""torch.nn.Linear vs the native linear layer on one LLM head shape; appends a row to a CSV
(model_name, in_features, out_features, batch, t_torch, t_native, ok) - seconds per call, the
format of the reference's my_lab/benchmark/benchmark_results.csv (my_lab/benchmark/matmul.py)."""
import argparse
import csv
import os

import torch

from layer import NativeLinear


def bench(fn, iters=50, warmup=10):
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--in_features", type=int, default=4096)
    ap.add_argument("--out_features", type=int, default=32768)
    ap.add_argument("--model_name", default="custom")
    ap.add_argument("--csv", default="benchmark_results.csv")
    args = ap.parse_args()
    torch.manual_seed(0)
    ref = torch.nn.Linear(args.in_features, args.out_features, device="cuda", dtype=torch.bfloat16)
    mine = NativeLinear.from_torch(ref)
    x = torch.randn(args.batch, args.in_features, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        y_ref, y = ref(x), mine(x)
        ok = bool((y.float() - y_ref.float()).abs().max() <= 2e-2 * y_ref.float().abs().max() + 1e-2)
        t_torch, t_native = bench(lambda: ref(x)), bench(lambda: mine(x))
    new = not os.path.exists(args.csv)
    with open(args.csv, "a", newline="") as f:
        wr = csv.writer(f)
        if new:
            wr.writerow(["model_name", "in_features", "out_features", "batch", "t_torch", "t_native", "ok"])
        wr.writerow([args.model_name, args.in_features, args.out_features, args.batch, f"{t_torch:.6f}", f"{t_native:.6f}", ok])
    gb = (args.in_features * args.out_features + args.batch * (args.in_features + args.out_features)) * 2 / 1e9
    print(f"{args.model_name} {args.in_features}->{args.out_features} batch {args.batch}: torch {t_torch * 1e6:.1f} us, "
          f"native {t_native * 1e6:.1f} us ({gb / t_native / 1e3:.2f} TB/s), ok={ok}", flush=True)


if __name__ == "__main__":
    main()
