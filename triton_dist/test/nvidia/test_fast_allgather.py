"""Low-latency all-gather of a small symmetric buffer, every protocol: correctness and latency.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_fast_allgather.py --mode push_2d_ll
    python triton_dist/test/nvidia/test_fast_allgather.py --virtual-ranks 8 --check

(CLI of reference python/triton_dist/test/nvidia/test_fast_allgather.py; the modes are the
reference's dispatch-table keys, kernels/nvidia/low_latency_allgather.py:957-965.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.nvidia.low_latency_allgather import create_fast_allgather_context, fast_allgather  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402

MODES = ["pull", "push_2d", "push_3d", "push_2d_ll", "push_2d_ll_multimem", "push_numa_2d", "push_numa_2d_ll",
         "push_numa_2d_ll_multinode"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="all", choices=MODES + ["all"])
    ap.add_argument("--nbytes", type=int, default=32768, help="bytes per rank")
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev = R.world_size, R.device
    n = args.nbytes // 2
    ctxs = R.collective(lambda r, env: create_fast_allgather_context(r, max_buffer_size=args.nbytes, env=env))
    bufs = R.collective(lambda r, env: env.heap.create_tensor([w, n], torch.bfloat16))
    for mode in (MODES if args.mode == "all" else [args.mode]):
        if args.check:
            for it in range(3):
                vals = {r: torch.randn(n, device=dev).bfloat16() for r in R.local_ranks}
                for r in R.local_ranks:
                    bufs[r][r].copy_(vals[r])
                R.barrier()
                R.run(lambda r, env: fast_allgather(bufs[r], ctxs[r], mode=mode))
                allv = R.all_gather(vals)
                for r in R.local_ranks:
                    ctxs[r].check_error()
                    for s in range(w):
                        assert torch.equal(bufs[r][s], allv[s]), f"mode {mode} rank {r} slot {s}"
            R.print(f"✅ fast_allgather mode={mode} nbytes/rank={args.nbytes} world={w} check passed")
        else:
            ms = R.time(lambda r, env: fast_allgather(bufs[r], ctxs[r], mode=mode), args.warmup, args.iters)
            R.print(f"fast_allgather {mode:>26}: {ms * 1e3:8.1f} us for {args.nbytes} B/rank")
    R.close()


if __name__ == "__main__":
    main()
