"""Ring put: every rank puts a block into its right neighbour's symmetric buffer with
putmem_signal and waits for the block from its left neighbour; repeated for w-1 hops the blocks
travel once around the ring (an all-gather built from the SHMEM primitives).

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_ring_put.py
    python triton_dist/test/nvidia/test_ring_put.py --virtual-ranks 4

(Scenario of reference python/triton_dist/test/nvidia/test_ring_put.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import triton_dist.language as dl  # noqa: E402
from triton_distributed_gh_amd.ops.common_ops import memcpy_async  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=65536, help="int32 elements per block")
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, n, dev = R.world_size, args.n, R.device
    bufs = R.collective(lambda r, env: env.heap.create_tensor([w, n], torch.int32))
    sigs = R.collective(lambda r, env: env.heap.create_tensor([w], torch.int32, arena="sig"))
    for r in R.local_ranks:
        bufs[r][r].fill_(r + 1)
    R.barrier()
    for hop in range(1, w):
        def step(r, env):
            right = (r + 1) % w
            blk = (r - hop + 1) % w                      # the block this rank forwards in this hop
            if env.backend == "host":
                env.heap.peer_view(bufs[r], right)[blk].copy_(bufs[r][blk])
            else:
                dst = env.heap.peer_ptr(bufs[r], right) + blk * n * 4
                memcpy_async(dst, bufs[r][blk], n * 4)
            dl.notify(sigs[r][blk:blk + 1], right, signal=hop, sig_op="set", comm_scope="intra_node", env=env)
            incoming = (r - hop) % w
            dl.wait(sigs[r][incoming:incoming + 1], 1, "sys", "acquire", waitValue=hop, env=env)
        R.run(step)
    R.barrier()
    for r in R.local_ranks:
        got = bufs[r][:, 0].tolist()
        assert got == [s + 1 for s in range(w)], f"rank {r}: {got}"
    R.print(f"✅ ring put ({w - 1} hops of {n * 4} bytes) on {w} ranks passed")
    R.close()


if __name__ == "__main__":
    main()
