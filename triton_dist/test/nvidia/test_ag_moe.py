"""AllGather + grouped (MoE) GEMM: out[ntokens*topk, N/w] = gather(A)[token] @ W[expert]^T.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_ag_moe.py --M 16384 --N 14336 --K 4096 --G 8 --topk 2
    python triton_dist/test/nvidia/test_ag_moe.py --M 1024 --N 1024 --K 512 --G 8 --topk 2 --virtual-ranks 4 --check

Defaults are the Mixtral-8x7B layer the reference benchmarks (E 8, topk 2, hidden 4096, ffn 14336).
(CLI of reference python/triton_dist/test/nvidia/test_ag_moe.py; golden = masked per-expert matmul, :39-55.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.nvidia.allgather_group_gemm import ag_group_gemm, create_ag_group_gemm_context  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=16384, help="tokens over all ranks")
    ap.add_argument("--N", type=int, default=14336)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--G", type=int, default=8, help="experts")
    ap.add_argument("--topk", type=int, default=2)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, dt = R.world_size, R.device, torch.bfloat16
    M, N, K, E, topk = args.M, args.N, args.K, args.G, args.topk
    ctxs = R.collective(lambda r, env: create_ag_group_gemm_context(M, N // w, K, E, topk, dt, r, w, env=env))
    g = torch.Generator(device="cpu").manual_seed(args.seed)
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(M)]).int().to(dev)   # same on every rank
    A = {r: (torch.rand(M // w, K, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    B = {r: (torch.rand(E, N // w, K, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    R.barrier()
    if args.check:
        for it in range(2):
            outs = R.run(lambda r, env: ag_group_gemm(A[r], B[r], ctxs[r], ids))
            a_full = torch.cat(R.all_gather(A)).float()
            flat = ids.reshape(-1).long()
            rows = a_full.repeat_interleave(topk, dim=0)
            for r in R.local_ranks:
                ctxs[r].check_error()
                ref = torch.empty(M * topk, N // w, device=dev)
                for e in range(E):
                    m = flat == e
                    ref[m] = rows[m] @ B[r][e].float().t()
                assert rel_err(outs[r], ref) < 1e-2, f"rank {r} iteration {it}"
        R.print(f"✅ ag_group_gemm M={M} N={N} K={K} E={E} topk={topk} world={w} check passed")
    else:
        ms = R.time(lambda r, env: ag_group_gemm(A[r], B[r], ctxs[r], ids), args.warmup, args.iters)
        tf = 2.0 * M * topk * (N // w) * K / (ms * 1e-3) / 1e12
        R.print_all(f"dist-triton ag_group_gemm #ranks {w}: {ms:.4f} ms ({tf:.1f} TFLOP/s per rank)")
    R.close()


if __name__ == "__main__":
    main()
