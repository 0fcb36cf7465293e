"""Bulk all-gather into a symmetric buffer, full-mesh push or pull: correctness and bandwidth.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_all_gather.py --nbytes 33554432
    python triton_dist/test/nvidia/test_all_gather.py --virtual-ranks 4 --check

xGMI is a full mesh, so the reference's ring / NUMA-2D methods (kernels/nvidia/allgather.py:140-260)
map onto the two full-mesh methods here.  (CLI of reference python/triton_dist/test/nvidia/test_all_gather.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.nvidia.allgather import (AllGatherMethod, all_gather_full_mesh,  # noqa: E402
                                                  create_fast_allgather_context, get_auto_all_gather_method)
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nbytes", type=int, default=1 << 20, help="bytes per rank")
    ap.add_argument("--method", default="auto", choices=["auto", "push", "pull"])
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev = R.world_size, R.device
    n = args.nbytes // 2
    method = {"auto": get_auto_all_gather_method(w, w), "push": AllGatherMethod.FullMeshPush,
              "pull": AllGatherMethod.FullMeshPull}[args.method]
    ctxs = R.collective(lambda r, env: create_fast_allgather_context(r, max_buffer_size=args.nbytes, env=env))
    bufs = R.collective(lambda r, env: env.heap.create_tensor([w, n], torch.bfloat16))
    vals = {r: torch.randn(n, device=dev).bfloat16() for r in R.local_ranks}
    R.barrier()
    if args.check:
        for it in range(3):
            R.run(lambda r, env: all_gather_full_mesh(vals[r], bufs[r], ctxs[r], method))
            allv = R.all_gather(vals)
            for r in R.local_ranks:
                ctxs[r].check_error()
                for s in range(w):
                    assert torch.equal(bufs[r][s], allv[s]), f"rank {r} slot {s} iteration {it}"
        R.print(f"✅ all_gather {method.name} nbytes/rank={args.nbytes} world={w} check passed")
    else:
        ms = R.time(lambda r, env: all_gather_full_mesh(vals[r], bufs[r], ctxs[r], method), args.warmup, args.iters)
        R.print_all(f"all_gather {method.name} #ranks {w}: {ms * 1e3:.1f} us (busbw {(w - 1) * args.nbytes / ms / 1e6:.1f} GB/s)")
    R.close()


if __name__ == "__main__":
    main()
