"""GEMM + ReduceScatter, intra-node: correctness (--check) or latency vs torch + RCCL.

    bash scripts/launch_amd.sh python/triton_dist/test/amd/test_gemm_rs_intra_node.py 8192 4096 12288
    python triton_dist/test/amd/test_gemm_rs_intra_node.py 2048 1024 4096 --virtual-ranks 4 --check

Positional M N K are GLOBAL: A is [M, K/w] per rank, B is [N, K/w] per rank; returns [M/w, N].
(CLI of reference python/triton_dist/test/amd/test_gemm_rs_intra_node.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402
import torch.distributed as dist

from triton_dist.kernels.amd import create_gemm_rs_intra_node_context, gemm_rs_intra_node
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err

DTYPES = {"bfloat16": torch.bfloat16, "float16": torch.float16}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("M", type=int)
    ap.add_argument("N", type=int)
    ap.add_argument("K", type=int)
    ap.add_argument("--dtype", default="bfloat16", choices=list(DTYPES))
    ap.add_argument("--verify-iters", default=5, type=int)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dt, dev = R.world_size, DTYPES[args.dtype], R.device
    M, N, K = args.M, args.N, args.K
    assert M % w == 0 and K % w == 0
    ctxs = R.collective(lambda r, env: create_gemm_rs_intra_node_context(M, N, dt, r, w, env.tp_group, env=env))

    def make():
        s = 0.01
        return ({r: ((torch.rand(M, K // w, device=dev) * 2 - 1) * s * (r + 1)).to(dt) for r in R.local_ranks},
                {r: ((torch.rand(N, K // w, device=dev) * 2 - 1)).to(dt) for r in R.local_ranks})

    if args.check:
        for it in range(args.verify_iters):
            A, B = make()
            R.barrier()
            out = R.run(lambda r, env: gemm_rs_intra_node(A[r], B[r], ctxs[r]))
            parts = R.all_gather({r: (A[r].float() @ B[r].float().t()).to(dt) for r in R.local_ranks})
            total = sum(p.float() for p in parts)
            for r in R.local_ranks:
                ctxs[r].check_error()
                err = rel_err(out[r], total[r * (M // w):(r + 1) * (M // w)])
                assert err < 1e-2, f"rank {r} iteration {it}: relative error {err}"
        R.print(f"✅ gemm_rs_intra_node M={M} N={N} K={K} world={w} check passed")
    else:
        A, B = make()
        t_dist = R.time(lambda r, env: gemm_rs_intra_node(A[r], B[r], ctxs[r]), args.warmup, args.iters)
        R.print_all(f"dist-triton #ranks {w}: {t_dist:.4f} ms")
        if not R.virtual:
            env = R.envs[0]
            full = torch.empty(M, N, device=dev, dtype=dt)
            out = torch.empty(M // w, N, device=dev, dtype=dt)

            def torch_gemm_rs(r, e):
                torch.matmul(A[r], B[r].t(), out=full)
                if w > 1:
                    dist.reduce_scatter_tensor(out, full, group=env.tp_group)
                return out

            t_torch = R.time(torch_gemm_rs, args.warmup, args.iters)
            R.print_all(f"torch #ranks {w}: {t_torch:.4f} ms  (speed-up {t_torch / t_dist:.3f}x)")
    R.close()


if __name__ == "__main__":
    main()
