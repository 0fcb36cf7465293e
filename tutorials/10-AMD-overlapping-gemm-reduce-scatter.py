"""Tutorial 10 - AMD entry points: create_gemm_rs_intra_node_context / gemm_rs_intra_node.

out[M/w, N] = ReduceScatter_M(A[M, K/w] @ B[N, K/w]^T).  ONE persistent kernel: tiles are produced in
destination order (the peer "after" this rank first, own segment last), the epilogue stores each tile
straight into the owning rank's symmetric buffer over xGMI, per-chunk counters raise an epoch flag on
the destination, and workgroups that run out of GEMM tiles reduce chunks whose w partials have
arrived - the reduction of early chunks overlaps the GEMM of late ones (csrc/ops/gemm_rs.hip).

    python tutorials/10-AMD-overlapping-gemm-reduce-scatter.py --virtual-ranks 4 --M 2048 --N 1024 --K 2048
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402,F401

from triton_dist.kernels.amd import create_gemm_rs_intra_node_context, gemm_rs_intra_node  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--N", type=int, default=8192)
    ap.add_argument("--K", type=int, default=8192)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, M, N, K = R.world_size, R.device, args.M, args.N, args.K
    dt = torch.bfloat16
    A = {r: (torch.rand(M, K // w, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    B = {r: (torch.rand(N, K // w, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    ctxs = R.collective(lambda r, env: create_gemm_rs_intra_node_context(M, N, dt, r, w, env.tp_group, env=env))
    R.barrier()
    out = R.run(lambda r, env: gemm_rs_intra_node(A[r], B[r], ctxs[r]))
    total = sum(p.float() for p in R.all_gather({r: (A[r].float() @ B[r].float().t()).to(dt) for r in R.local_ranks}))
    for r in R.local_ranks:
        assert rel_err(out[r], total[r * (M // w):(r + 1) * (M // w)]) < 1e-2
    t = R.time(lambda r, env: gemm_rs_intra_node(A[r], B[r], ctxs[r]), args.warmup, args.iters)
    R.print(f"✅ tutorial 10: gemm_rs_intra_node M={M} N={N} K={K} on {w} ranks correct, {t:.3f} ms per call")
    R.close()


if __name__ == "__main__":
    main()
