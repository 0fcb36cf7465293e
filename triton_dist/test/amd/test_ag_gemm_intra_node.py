"""AllGather + GEMM, intra-node: correctness (--check) or latency vs torch + RCCL.

    bash scripts/launch_amd.sh python/triton_dist/test/amd/test_ag_gemm_intra_node.py 8192 11008 4096
    python triton_dist/test/amd/test_ag_gemm_intra_node.py 2048 4096 1024 --virtual-ranks 4 --check

Positional M N K are GLOBAL sizes: A is [M, K] row-sharded, B is [N, K] row-sharded (N/w per rank);
the op returns [M, N/w] per rank.  (CLI of reference python/triton_dist/test/amd/test_ag_gemm_intra_node.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402
import torch.distributed as dist

from triton_dist.kernels.amd import ag_gemm_intra_node, create_ag_gemm_intra_node_context
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err

DTYPES = {"bfloat16": torch.bfloat16, "float16": torch.float16}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("M", type=int)
    ap.add_argument("N", type=int)
    ap.add_argument("K", type=int)
    ap.add_argument("--dtype", default="bfloat16", choices=list(DTYPES))
    ap.add_argument("--chunk_m", default=256, type=int, help="rows per all-gather chunk / flag")
    ap.add_argument("--producer", default=None, choices=[None, "kernel", "copy_engine"])
    ap.add_argument("--verify-iters", default=5, type=int)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dt, dev = R.world_size, DTYPES[args.dtype], R.device
    M, N, K = args.M, args.N, args.K
    assert M % w == 0 and N % w == 0
    ctxs = R.collective(lambda r, env: create_ag_gemm_intra_node_context(
        M, N // w, K, dt, dt, r, w, env.tp_group, M_PER_CHUNK=args.chunk_m, producer=args.producer, env=env))

    def make():
        return ({r: (torch.rand(M // w, K, device=dev) * 2 - 1).to(dt) for r in R.local_ranks},
                {r: (torch.rand(N // w, K, device=dev) * 2 - 1).to(dt) for r in R.local_ranks})

    if args.check:
        for it in range(args.verify_iters):
            A, B = make()
            R.barrier()
            out = R.run(lambda r, env: ag_gemm_intra_node(A[r], B[r], False, ctxs[r]))
            full = torch.cat(R.all_gather(A)).float()
            for r in R.local_ranks:
                ctxs[r].check_error()
                err = rel_err(out[r], full @ B[r].float().t())
                assert err < 1e-2, f"rank {r} iteration {it}: relative error {err}"
        R.print(f"✅ ag_gemm_intra_node M={M} N={N} K={K} world={w} check passed")
    else:
        A, B = make()
        t_dist = R.time(lambda r, env: ag_gemm_intra_node(A[r], B[r], False, ctxs[r]), args.warmup, args.iters)
        R.print_all(f"dist-triton #ranks {w}: {t_dist:.4f} ms")
        if not R.virtual:
            env = R.envs[0]
            full = torch.empty(M, K, device=dev, dtype=dt)

            def torch_ag_gemm(r, e):
                if w > 1:
                    dist.all_gather_into_tensor(full, A[r], group=env.tp_group)
                    return torch.matmul(full, B[r].t())
                return torch.matmul(A[r], B[r].t())

            t_torch = R.time(torch_ag_gemm, args.warmup, args.iters)
            R.print_all(f"torch #ranks {w}: {t_torch:.4f} ms  (speed-up {t_torch / t_dist:.3f}x)")
    R.close()


if __name__ == "__main__":
    main()
