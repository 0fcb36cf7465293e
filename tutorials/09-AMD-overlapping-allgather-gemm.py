"""Tutorial 09 - AMD entry points: create_ag_gemm_intra_node_context / ag_gemm_intra_node.

C[M, N/w] = AllGather_M(A[M/w, K]) @ B[N/w, K]^T.  A push kernel writes every rank's rows into the
peers' symmetric workspaces chunk by chunk and raises one flag per (source rank, chunk); the
persistent MFMA GEMM pulls tile jobs in expected-arrival order (own rows first) and waits only for the
flags its tile needs - compute starts immediately and the transfer hides behind it
(csrc/ops/ag_gemm.hip).  Calls are numbered by an epoch and buffers alternate, so no reset or barrier
follows the GEMM.

    python tutorials/09-AMD-overlapping-allgather-gemm.py --virtual-ranks 4 --M 2048 --N 2048 --K 1024
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402,F401

from triton_dist.kernels.amd import ag_gemm_intra_node, create_ag_gemm_intra_node_context  # noqa: E402


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
    A = {r: (torch.rand(M // w, K, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    B = {r: (torch.rand(N // w, K, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    ctxs = R.collective(lambda r, env: create_ag_gemm_intra_node_context(M, N // w, K, dt, dt, r, w, env.tp_group, env=env))
    R.barrier()
    out = R.run(lambda r, env: ag_gemm_intra_node(A[r], B[r], False, ctxs[r]))
    full = torch.cat(R.all_gather(A)).float()
    for r in R.local_ranks:
        assert rel_err(out[r], full @ B[r].float().t()) < 1e-2
    t = R.time(lambda r, env: ag_gemm_intra_node(A[r], B[r], False, ctxs[r]), args.warmup, args.iters)
    R.print(f"✅ tutorial 09: ag_gemm_intra_node M={M} N={N} K={K} on {w} ranks correct, {t:.3f} ms per call")
    R.close()


if __name__ == "__main__":
    main()
