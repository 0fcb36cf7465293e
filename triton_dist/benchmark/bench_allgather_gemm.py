"""AG-GEMM benchmark over LLM layer shapes: fused op vs all-gather only vs GEMM only vs torch + RCCL.

    bash scripts/launch_amd.sh python/triton_dist/benchmark/bench_allgather_gemm.py --M 8192 --dump_csv ag_gemm.csv

(Reference: python/triton_dist/benchmark/bench_allgather_gemm.py:174-181 layer_configs; reports
triton / ag_only / gemm_only / torch in ms.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
os.environ.setdefault("RANDOM_PARAMS", "1")

import torch  # noqa: E402

import torch.distributed as dist  # noqa: E402

from triton_dist.kernels.amd import ag_gemm_intra_node, create_ag_gemm_intra_node_context  # noqa: E402
from triton_distributed_gh_amd.ops.gemm import gemm_nt  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402

LAYER_CONFIGS = {                       # (N, K) of the column-parallel projection
    "LLaMA-7B": (11008 * 2, 4096), "LLaMA-3.1-8B": (14336 * 2, 4096), "LLaMA-3.1-70B": (28672 * 2, 8192),
    "LLaMA-3.1-405B": (53248 * 2, 16384), "Mistral-7B": (14336 * 2, 4096), "Qwen2-72B": (29568 * 2, 8192),
    "Qwen3-32B": (25600 * 2, 5120), "north-star": (8192, 8192),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=8192)
    ap.add_argument("--models", nargs="*", default=list(LAYER_CONFIGS))
    ap.add_argument("--dump_csv", default=None)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, dt, M = R.world_size, R.device, torch.bfloat16, args.M
    rows = []
    for name in args.models:
        N, K = LAYER_CONFIGS[name]
        if N % (w * 8) or M % w:
            continue
        ctxs = R.collective(lambda r, env: create_ag_gemm_intra_node_context(M, N // w, K, dt, dt, r, w, env.tp_group, env=env))
        A = {r: (torch.rand(M // w, K, device=dev) * 2 - 1).to(dt) for r in R.local_ranks}
        B = {r: (torch.rand(N // w, K, device=dev) * 2 - 1).to(dt) for r in R.local_ranks}
        full = {r: torch.empty(M, K, device=dev, dtype=dt) for r in R.local_ranks}
        t_fused = R.time(lambda r, e: ag_gemm_intra_node(A[r], B[r], False, ctxs[r]), args.warmup, args.iters)
        t_gemm = R.time(lambda r, e: gemm_nt(full[r], B[r]), args.warmup, args.iters)
        t_ag = t_torch = float("nan")
        if not R.virtual and w > 1:
            g = R.envs[0].tp_group
            t_ag = R.time(lambda r, e: dist.all_gather_into_tensor(full[r], A[r], group=g), args.warmup, args.iters)

            def torch_fn(r, e):
                dist.all_gather_into_tensor(full[r], A[r], group=g)
                return torch.matmul(full[r], B[r].t())

            t_torch = R.time(torch_fn, args.warmup, args.iters)
        rows.append((name, M, N, K, w, t_fused, t_ag, t_gemm, t_torch))
        R.print(f"{name:16s} M={M} N={N} K={K} tp={w}: fused {t_fused:.3f} ms | ag_only {t_ag:.3f} | gemm_only {t_gemm:.3f} | "
                f"torch {t_torch:.3f}")
        for c in ctxs.values():
            for parity in c.workspaces:
                R.envs[c.rank if R.virtual else 0].heap.free_tensor(parity[c.rank])
    if args.dump_csv and R.is_main:
        with open(args.dump_csv, "w") as f:
            f.write("model,M,N,K,tp,fused_ms,ag_only_ms,gemm_only_ms,torch_ms\n")
            for row in rows:
                f.write(",".join(str(x) for x in row) + "\n")
    R.close()


if __name__ == "__main__":
    main()
