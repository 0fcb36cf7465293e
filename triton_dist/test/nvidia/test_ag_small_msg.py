"""Latency of the low-latency all-gather protocols over small messages (the decode-time path:
flash-decode partials, router metadata).

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_ag_small_msg.py
    python triton_dist/test/nvidia/test_ag_small_msg.py --virtual-ranks 4 --check

Sweeps bytes-per-rank from --min-nbytes to --max-nbytes (x4 steps) over the pull, push and LL
protocols.  (CLI of reference python/triton_dist/test/nvidia/test_ag_small_msg.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.nvidia.low_latency_allgather import create_fast_allgather_context, fast_allgather  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402

MODES = ["pull", "push_2d", "push_2d_ll"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-nbytes", type=int, default=1024)
    ap.add_argument("--max-nbytes", type=int, default=262144)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev = R.world_size, R.device
    ctxs = R.collective(lambda r, env: create_fast_allgather_context(r, max_buffer_size=args.max_nbytes, env=env))
    nbytes = args.min_nbytes
    while nbytes <= args.max_nbytes:
        n = nbytes // 2
        bufs = R.collective(lambda r, env: env.heap.create_tensor([w, n], torch.bfloat16))
        for mode in MODES:
            if args.check:
                vals = {r: torch.randn(n, device=dev).bfloat16() for r in R.local_ranks}
                for r in R.local_ranks:
                    bufs[r][r].copy_(vals[r])
                R.barrier()
                R.run(lambda r, env: fast_allgather(bufs[r], ctxs[r], mode=mode))
                allv = R.all_gather(vals)
                for r in R.local_ranks:
                    ctxs[r].check_error()
                    for s in range(w):
                        assert torch.equal(bufs[r][s], allv[s]), f"mode {mode} nbytes {nbytes} rank {r} slot {s}"
            else:
                ms = R.time(lambda r, env: fast_allgather(bufs[r], ctxs[r], mode=mode), args.warmup, args.iters)
                R.print(f"fast_allgather {mode:>10} {nbytes:>8} B/rank: {ms * 1e3:8.1f} us")
        nbytes *= 4
    if args.check:
        R.print(f"✅ small-message all-gather ({', '.join(MODES)}) world={w} check passed")
    R.close()


if __name__ == "__main__":
    main()
