"""Tutorial 03 - "inter-node" all-gather entry points on a single-node framework.

The reference's tutorial 03 builds a 2-D all-gather (intra-node ring + inter-node NVSHMEM puts).  This
framework targets ONE xGMI node, where every peer is one hop away; the multi-node method names of
the all-gather layer are kept so reference code runs unchanged, and all of them resolve to the
full-mesh kernels.  Splitting the node into "virtual nodes" (nnodes > 1) only changes bookkeeping.

    python tutorials/03-inter-node-allgather.py --virtual-ranks 4
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402,F401

from triton_dist.layers.nvidia.low_latency_allgather_layer import AllGatherLayer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev = R.world_size, R.device
    nbytes = 64 * 1024
    layers = R.collective(lambda r, env: AllGatherLayer(nnodes=2 if w % 2 == 0 else 1, world_size=w, rank=r,
                                                       max_buffer_size=nbytes, env=env))
    sym = R.collective(lambda r, env: env.heap.create_tensor([w, nbytes], torch.int8))
    want = torch.arange(w, dtype=torch.int8)[:, None].expand(w, nbytes)
    for method in ("forward_pull", "forward_push_2d", "forward_push_numa_2d", "forward_push_2d_ll",
                   "forward_push_numa_2d_ll_multinode"):
        for r in R.local_ranks:
            sym[r].zero_()
            sym[r][r].fill_(r)
        R.barrier()
        R.run(lambda r, env: getattr(layers[r], method)(sym[r]))
        for r in R.local_ranks:
            assert torch.equal(sym[r].cpu(), want), method
        R.print(f"{method} ok")
    R.print("✅ tutorial 03 passed")
    R.close()


if __name__ == "__main__":
    main()
