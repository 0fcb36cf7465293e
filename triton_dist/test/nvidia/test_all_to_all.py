"""Low-latency all-to-all (EP dispatch of expert-sorted token rows, optional fp8 + scales).

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_all_to_all.py -M 128 -N 7168 --topk 8 -G 32 --with-scale
    python triton_dist/test/nvidia/test_all_to_all.py --virtual-ranks 4 --check

-M tokens per rank, -N hidden, -G experts per rank; every token picks --topk experts at random, so
each rank sends M*topk rows sorted by destination expert (the reference's DeepSeek-style shape:
128 tokens/rank, topk 8, hidden 7168, README.md:97).  Payload compared bit-exact with --check.
(CLI of reference python/triton_dist/test/nvidia/test_all_to_all.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.nvidia.low_latency_all_to_all import (all_to_all_post_process, create_all_to_all_context,  # noqa: E402
                                                               fast_all_to_all, quantize_fp8_groupwise)
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-M", type=int, default=128)
    ap.add_argument("-N", type=int, default=7168)
    ap.add_argument("-G", type=int, default=8, help="experts per rank")
    ap.add_argument("--topk", type=int, default=8)
    ap.add_argument("--with-scale", action="store_true", help="send fp8 e4m3 rows + per-128 fp32 scales")
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev = R.world_size, R.device
    E, rows = w * args.G, args.M * args.topk
    dt = torch.float8_e4m3fn if args.with_scale else torch.bfloat16
    ctxs = R.collective(lambda r, env: create_all_to_all_context(rows, args.N, r, E, w, args.G, dt, env=env))

    def make(r):
        g = torch.Generator(device="cpu").manual_seed(args.seed + r)
        dest = torch.sort(torch.randint(0, E, (rows,), generator=g)).values
        cum = torch.zeros(E + 1, dtype=torch.int32)
        cum[1:] = torch.cumsum(torch.bincount(dest, minlength=E), 0)
        x = torch.randn(rows, args.N, generator=g).to(dev).bfloat16()
        if args.with_scale:
            q, s = quantize_fp8_groupwise(x)
            return q, cum.to(dev), s
        return x, cum.to(dev), None

    data = {r: make(r) for r in R.local_ranks}
    if args.check:
        for it in range(3):
            outs = R.run(lambda r, env: fast_all_to_all(ctxs[r], *data[r]))
            sends = R.all_gather({r: data[r][0].view(torch.uint8) for r in R.local_ranks})
            cums = R.all_gather({r: data[r][1] for r in R.local_ranks})
            for r in R.local_ranks:
                ctxs[r].check_error()
                splits, recv, rscale = outs[r]
                for s in range(w):
                    st, ed = int(cums[s][r * args.G]), int(cums[s][(r + 1) * args.G])
                    assert torch.equal(recv[s, :ed - st].view(torch.uint8), sends[s][st:ed]), f"rank {r} src {s}"
                toks, _, per_exp = all_to_all_post_process(ctxs[r], splits, recv, rscale)
                assert toks.shape[0] == int(per_exp.sum())
        R.print(f"✅ fast_all_to_all M={args.M} hidden={args.N} topk={args.topk} world={w} check passed")
    else:
        ms = R.time(lambda r, env: fast_all_to_all(ctxs[r], *data[r]), args.warmup, args.iters)
        nbytes = rows * args.N * data[R.local_ranks[0]][0].element_size()
        R.print_all(f"fast_all_to_all #ranks {w}: {ms * 1e3:.1f} us ({nbytes / ms / 1e6:.1f} GB/s sent per rank)")
    R.close()


if __name__ == "__main__":
    main()
