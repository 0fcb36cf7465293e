"""MoE grouped GEMM + top-k reduce + reduce-scatter: out[ntokens/w, hidden].

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_moe_reduce_rs.py 16384 4096 14336 8 2
    python triton_dist/test/nvidia/test_moe_reduce_rs.py 512 1024 1024 8 2 --virtual-ranks 4 --check

Positional: M (tokens over all ranks) N (hidden) K (ffn, split over ranks) E topk - the reference's
Mixtral shape is 16384 4096 14336 8 2.  x is [M*topk, K/w] per rank, expert weights [E, N, K/w].
(CLI of reference python/triton_dist/test/nvidia/test_moe_reduce_rs.py; golden :84-103.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.nvidia.moe_reduce_rs import create_moe_rs_context, run_moe_reduce_rs  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    for name in ("M", "N", "K", "E", "topk"):
        ap.add_argument(name, type=int)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, dt = R.world_size, R.device, torch.bfloat16
    M, H, K, E, topk = args.M, args.N, args.K, args.E, args.topk
    ctxs = R.collective(lambda r, env: create_moe_rs_context(r, w, w, M, H, E, topk, dt, env=env))
    g = torch.Generator(device="cpu").manual_seed(args.seed)
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(M)]).int().to(dev)
    rw = torch.rand(M, topk, generator=g).to(dev)
    X = {r: (torch.rand(M * topk, K // w, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    W = {r: (torch.rand(E, H, K // w, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    R.barrier()
    if args.check:
        outs = R.run(lambda r, env: run_moe_reduce_rs(X[r], W[r], ids, rw, ctxs[r]))
        flat = ids.reshape(-1).long()
        parts = {}
        for r in R.local_ranks:
            part = torch.empty(M * topk, H, device=dev)
            for e in range(E):
                m = flat == e
                part[m] = X[r][m].float() @ W[r][e].float().t()
            part = (part * rw.reshape(-1, 1)).to(dt).float().view(M, topk, H).sum(1)
            parts[r] = part.to(dt) if w > 1 else part
        total = sum(p.float() for p in R.all_gather(parts))
        for r in R.local_ranks:
            ctxs[r].check_error()
            assert rel_err(outs[r], total[r * (M // w):(r + 1) * (M // w)]) < 2e-2, f"rank {r}"
        R.print(f"✅ moe_reduce_rs M={M} hidden={H} K={K} E={E} topk={topk} world={w} check passed")
    else:
        ms = R.time(lambda r, env: run_moe_reduce_rs(X[r], W[r], ids, rw, ctxs[r]), args.warmup, args.iters)
        tf = 2.0 * M * topk * H * (K // w) / (ms * 1e-3) / 1e12
        R.print_all(f"dist-triton moe_reduce_rs #ranks {w}: {ms:.4f} ms ({tf:.1f} TFLOP/s per rank)")
    R.close()


if __name__ == "__main__":
    main()
