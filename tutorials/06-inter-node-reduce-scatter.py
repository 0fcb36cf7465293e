"""Tutorial 06 - the "inter-node" reduce-scatter entry points.

Every rank publishes its [M, N] partial in a symmetric buffer and flags its peers; each rank then
reads ITS row segment from all w buffers over w-1 different xGMI links at once and adds in fp32
(csrc/ops/reduce_scatter.hip).  The reference's tutorial 05 scatters with the copy engine and then
ring-reduces; on a full mesh the ring only adds hops.

    python tutorials/06-inter-node-reduce-scatter.py --virtual-ranks 4
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402,F401

from triton_dist.kernels.nvidia.reduce_scatter import create_reduce_scater_2d_ctx, reduce_scatter_multi_node  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=2048)
    ap.add_argument("--N", type=int, default=1024)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, M, N = R.world_size, R.device, args.M, args.N
    dt = torch.bfloat16 if dev.type == "cuda" else torch.float32
    ctxs = R.collective(lambda r, env: create_reduce_scater_2d_ctx(M, N, r, w, w, dt, env=env))
    x = {r: (torch.rand(M, N, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    R.barrier()
    out = R.run(lambda r, env: reduce_scatter_multi_node(x[r], ctxs[r]))
    total = sum(t.float() for t in R.all_gather(x))
    for r in R.local_ranks:
        assert rel_err(out[r], total[r * (M // w):(r + 1) * (M // w)]) < 2e-2
    R.print(f"✅ tutorial 06: reduce_scatter_multi_node (one xGMI node: same one-hop kernel) of [{M}, {N}] over {w} ranks matches the fp32 sum")
    R.close()


if __name__ == "__main__":
    main()
