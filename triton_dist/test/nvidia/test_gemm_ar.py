"""GEMM + AllReduce fused op: correctness (--check) or latency vs torch.matmul + RCCL all_reduce.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_gemm_ar.py 128 8192 8192 --low-latency
    python triton_dist/test/nvidia/test_gemm_ar.py 128 2048 1024 --virtual-ranks 4 --check

Positional M N K: A is [M, K/w] per rank, B is [N, K/w] per rank, every rank gets the full [M, N].
(CLI of reference python/triton_dist/test/nvidia/test_gemm_ar.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import torch.distributed as dist  # noqa: E402

from triton_dist.kernels.nvidia.gemm_allreduce import (create_gemm_ar_context, create_ll_gemm_ar_context,  # noqa: E402
                                                       gemm_allreduce_op, low_latency_gemm_allreduce_op)
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("M", type=int)
    ap.add_argument("N", type=int)
    ap.add_argument("K", type=int)
    ap.add_argument("--low-latency", action="store_true")
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, dt = R.world_size, R.device, torch.bfloat16
    M, N, K = args.M, args.N, args.K
    op = low_latency_gemm_allreduce_op if args.low_latency else gemm_allreduce_op
    if args.low_latency:
        ctxs = R.collective(lambda r, env: create_ll_gemm_ar_context(r, w, w, M, N, dt, env=env))
    else:
        ctxs = R.collective(lambda r, env: create_gemm_ar_context(None, r, w, w, M, N, dt, env=env))
    A = {r: (torch.rand(M, K // w, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    B = {r: (torch.rand(N, K // w, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    if args.check:
        for it in range(3):
            outs = R.run(lambda r, env: op(ctxs[r], A[r], B[r]))
            parts = R.all_gather({r: (A[r].float() @ B[r].float().t()).to(dt) for r in R.local_ranks})
            ref = sum(p.float() for p in parts)
            for r in R.local_ranks:
                ctxs[r].check_error()
                assert rel_err(outs[r], ref) < 1e-2, f"rank {r} iteration {it}"
        R.print(f"✅ gemm_allreduce M={M} N={N} K={K} world={w} check passed")
    else:
        ms = R.time(lambda r, env: op(ctxs[r], A[r], B[r]), args.warmup, args.iters)
        R.print_all(f"dist-triton #ranks {w}: {ms:.4f} ms")
        if not R.virtual:
            env = R.envs[0]

            def torch_gemm_ar(r, e):
                c = torch.matmul(A[r], B[r].t())
                if w > 1:
                    dist.all_reduce(c, group=env.tp_group)
                return c

            ms_t = R.time(torch_gemm_ar, args.warmup, args.iters)
            R.print_all(f"torch #ranks {w}: {ms_t:.4f} ms  (speed-up {ms_t / ms:.3f}x)")
    R.close()


if __name__ == "__main__":
    main()
