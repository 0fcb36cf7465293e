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

# the reference's DTYPE_MAP (test_ag_gemm_intra_node.py:122-129); 8-bit operands give a bf16 result here
DTYPES = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float8_e4m3fn": torch.float8_e4m3fn,
          "float8_e5m2": torch.float8_e5m2, "s8": torch.int8}


def rand_operand(rows, cols, dt, dev):
    if dt == torch.int8:
        return torch.randint(-127, 128, (rows, cols), device=dev, dtype=torch.int8)
    return (torch.rand(rows, cols, device=dev) * 2 - 1).to(dt)


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
        return ({r: rand_operand(M // w, K, dt, dev) for r in R.local_ranks},
                {r: rand_operand(N // w, K, dt, dev) for r in R.local_ranks})

    if args.check:
        for it in range(args.verify_iters):
            A, B = make()
            R.barrier()
            out = R.run(lambda r, env: ag_gemm_intra_node(A[r], B[r], False, ctxs[r]))
            # (8-bit tensors travel through the CPU all-gather as bytes)
            gathered = R.all_gather({r: A[r].view(torch.uint8) if dt.itemsize == 1 else A[r] for r in A})
            full = torch.cat([g.view(dt) if dt.itemsize == 1 else g for g in gathered]).float()
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
            # the library has no fp8 / int8 GEMM of this layout to compare with: its baseline multiplies in bf16
            mm_dt = torch.bfloat16 if dt.itemsize == 1 else dt
            Bm = {r: B[r].to(mm_dt) for r in B}

            def torch_ag_gemm(r, e):
                if w > 1:
                    dist.all_gather_into_tensor(full.view(torch.uint8) if dt.itemsize == 1 else full,
                                                A[r].view(torch.uint8) if dt.itemsize == 1 else A[r], group=env.tp_group)
                    return torch.matmul(full.to(mm_dt), Bm[r].t())
                return torch.matmul(A[r].to(mm_dt), Bm[r].t())

            t_torch = R.time(torch_ag_gemm, args.warmup, args.iters)
            R.print_all(f"torch #ranks {w}: {t_torch:.4f} ms  (speed-up {t_torch / t_dist:.3f}x)")
    R.close()


if __name__ == "__main__":
    main()
