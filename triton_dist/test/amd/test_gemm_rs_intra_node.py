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

# the reference's DTYPE_MAP (test_gemm_rs_intra_node.py); 8-bit operands give bf16 partials and a bf16 result here
DTYPES = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float8_e4m3fn": torch.float8_e4m3fn,
          "float8_e5m2": torch.float8_e5m2, "s8": torch.int8}


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
    odt = torch.bfloat16 if dt.itemsize == 1 else dt          # type of the partials and of the result
    ctxs = R.collective(lambda r, env: create_gemm_rs_intra_node_context(M, N, odt, r, w, env.tp_group, env=env))
    # int8 products are de-quantised by a per-row scale so that the bf16 partials stay in range
    sa = torch.full((M,), 1e-4, device=dev) if dt == torch.int8 else None

    def make():
        if dt == torch.int8:
            return ({r: torch.randint(-127, 128, (M, K // w), device=dev, dtype=dt) for r in R.local_ranks},
                    {r: torch.randint(-127, 128, (N, K // w), device=dev, dtype=dt) for r in R.local_ranks})
        s = 0.01 if dt.itemsize == 2 else 1.0
        return ({r: ((torch.rand(M, K // w, device=dev) * 2 - 1) * s * (r + 1)).to(dt) for r in R.local_ranks},
                {r: ((torch.rand(N, K // w, device=dev) * 2 - 1)).to(dt) for r in R.local_ranks})

    if args.check:
        for it in range(args.verify_iters):
            A, B = make()
            R.barrier()
            out = R.run(lambda r, env: gemm_rs_intra_node(A[r], B[r], ctxs[r], scale_a=sa))
            parts = R.all_gather({r: ((A[r].float() @ B[r].float().t()) * (sa[:, None] if sa is not None else 1.0)).to(odt)
                                  for r in R.local_ranks})
            total = sum(p.float() for p in parts)
            for r in R.local_ranks:
                ctxs[r].check_error()
                err = rel_err(out[r], total[r * (M // w):(r + 1) * (M // w)])
                assert err < 1e-2, f"rank {r} iteration {it}: relative error {err}"
        R.print(f"✅ gemm_rs_intra_node M={M} N={N} K={K} world={w} check passed")
    else:
        A, B = make()
        t_dist = R.time(lambda r, env: gemm_rs_intra_node(A[r], B[r], ctxs[r], scale_a=sa), args.warmup, args.iters)
        R.print_all(f"dist-triton #ranks {w}: {t_dist:.4f} ms")
        if not R.virtual:
            env = R.envs[0]
            full = torch.empty(M, N, device=dev, dtype=odt)
            out = torch.empty(M // w, N, device=dev, dtype=odt)
            # (the library baseline multiplies 8-bit operands as bf16)
            Am, Bm = {r: A[r].to(odt) for r in A}, {r: B[r].to(odt) for r in B}

            def torch_gemm_rs(r, e):
                torch.matmul(Am[r], Bm[r].t(), out=full)
                if w > 1:
                    dist.reduce_scatter_tensor(out, full, group=env.tp_group)
                return out

            t_torch = R.time(torch_gemm_rs, args.warmup, args.iters)
            R.print_all(f"torch #ranks {w}: {t_torch:.4f} ms  (speed-up {t_torch / t_dist:.3f}x)")
    R.close()


if __name__ == "__main__":
    main()
