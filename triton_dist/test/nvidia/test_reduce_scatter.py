"""Stand-alone 2D reduce-scatter: out[M/w, N] = sum over ranks of input[rank-th M slice].

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_reduce_scatter.py 8192 8192
    python triton_dist/test/nvidia/test_reduce_scatter.py 1024 512 --virtual-ranks 4 --check

(CLI of reference python/triton_dist/test/nvidia/test_reduce_scatter.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import torch.distributed as dist  # noqa: E402

from triton_dist.kernels.nvidia.reduce_scatter import create_reduce_scater_2d_ctx, reduce_scatter_2d_op  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("M", type=int)
    ap.add_argument("N", type=int)
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float16", "float32"])
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, dt = R.world_size, R.device, getattr(torch, args.dtype)
    M, N = args.M, args.N
    ctxs = R.collective(lambda r, env: create_reduce_scater_2d_ctx(M, N, r, w, w, dt, env=env))
    xs = {r: (torch.rand(M, N, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    R.barrier()
    if args.check:
        for it in range(3):
            outs = R.run(lambda r, env: reduce_scatter_2d_op(xs[r], ctxs[r]))
            total = sum(x.float() for x in R.all_gather(xs))
            for r in R.local_ranks:
                ctxs[r].check_error()
                assert rel_err(outs[r], total[r * (M // w):(r + 1) * (M // w)]) < 1e-2, f"rank {r} iteration {it}"
        R.print(f"✅ reduce_scatter_2d M={M} N={N} world={w} check passed")
    else:
        ms = R.time(lambda r, env: reduce_scatter_2d_op(xs[r], ctxs[r]), args.warmup, args.iters)
        nbytes = M * N * xs[R.local_ranks[0]].element_size()
        R.print_all(f"dist-triton reduce_scatter #ranks {w}: {ms:.4f} ms (busbw {(w - 1) / w * nbytes / ms / 1e6:.1f} GB/s)")
        if not R.virtual and w > 1:
            env = R.envs[0]
            out = torch.empty(M // w, N, device=dev, dtype=dt)
            ms_t = R.time(lambda r, e: dist.reduce_scatter_tensor(out, xs[r], group=env.tp_group), args.warmup, args.iters)
            R.print_all(f"torch reduce_scatter #ranks {w}: {ms_t:.4f} ms")
    R.close()


if __name__ == "__main__":
    main()
