"""AllReduce methods over the symmetric heap: correctness (--check) and latency per message size.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_allreduce.py --method two_shot
    python triton_dist/test/nvidia/test_allreduce.py --virtual-ranks 4 --check

(CLI of reference python/triton_dist/test/nvidia/test_allreduce.py: --method, --nbytes sweep, --check.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.nvidia.allreduce import (AllReduceMethod, all_reduce, create_allreduce_ctx,  # noqa: E402
                                                  get_auto_allreduce_method, to_allreduce_method)
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402

DTYPES = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--method", default="auto", help="auto | one_shot | two_shot | double_tree | ... (reference enum names)")
    ap.add_argument("--dtype", default="bfloat16", choices=list(DTYPES))
    ap.add_argument("--min-nbytes", type=int, default=4096)
    ap.add_argument("--max-nbytes", type=int, default=16 << 20)
    ap.add_argument("--stragglers", action="store_true", help="delay one rank in every other iteration")
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dt, dev = R.world_size, DTYPES[args.dtype], R.device
    ctxs = R.collective(lambda r, env: create_allreduce_ctx(args.max_nbytes, r, w, env=env))
    nbytes = args.min_nbytes
    while nbytes <= args.max_nbytes:
        n = nbytes // torch.empty((), dtype=dt).element_size()
        method = get_auto_allreduce_method(nbytes) if args.method == "auto" else to_allreduce_method(args.method)
        xs = {r: (torch.rand(n, device=dev) - 0.5).to(dt) for r in R.local_ranks}
        if args.check:
            for it in range(3):
                opt = (w - 1, 2_000_000) if (args.stragglers and it % 2) else None
                outs = R.run(lambda r, env: all_reduce(xs[r], method, ctxs[r], straggler_option=opt))
                ref = sum(x.float() for x in R.all_gather(xs))
                for r in R.local_ranks:
                    ctxs[r].check_error()
                    assert torch.allclose(outs[r].float(), ref, atol=2e-2 * w, rtol=2e-2), f"rank {r} nbytes {nbytes}"
            R.print(f"✅ all_reduce {method.name} nbytes={nbytes} world={w} check passed")
        else:
            ms = R.time(lambda r, env: all_reduce(xs[r], method, ctxs[r]), args.warmup, args.iters)
            bus = 2 * (w - 1) / w * nbytes / (ms * 1e-3) / 1e9
            R.print(f"all_reduce {method.name:>12} nbytes {nbytes:>10}: {ms * 1e3:9.1f} us  busbw {bus:7.1f} GB/s")
        nbytes *= 4
    R.close()


if __name__ == "__main__":
    main()
