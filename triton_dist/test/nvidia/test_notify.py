"""dl.notify across ranks: every rank raises a word on every peer with "set", then bumps a counter
word on every peer with "add"; values are checked after a barrier.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_notify.py
    python triton_dist/test/nvidia/test_notify.py --virtual-ranks 4

(Scenario of reference python/triton_dist/test/nvidia/test_notify.py: notify with sig_op set / add
and comm_scope intra_node, checked against the expected per-rank values.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import triton_dist.language as dl  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=10)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w = R.world_size
    sets = R.collective(lambda r, env: env.heap.create_tensor([w], torch.int32, arena="sig"))
    adds = R.collective(lambda r, env: env.heap.create_tensor([1], torch.int32, arena="sig"))
    R.barrier()
    for rnd in range(1, args.rounds + 1):
        def step(r, env):
            for peer in range(w):
                dl.notify(sets[r][r:r + 1], peer, signal=1000 * rnd + r, sig_op="set", comm_scope="intra_node", env=env)
                dl.notify(adds[r], peer, signal=r + 1, sig_op="add", comm_scope="intra_node", env=env)
        R.run(step)
        R.barrier()
        for r in R.local_ranks:
            assert sets[r].tolist() == [1000 * rnd + s for s in range(w)], f"rank {r} round {rnd}: {sets[r].tolist()}"
            assert int(adds[r][0]) == rnd * w * (w + 1) // 2, f"rank {r} round {rnd}: {int(adds[r][0])}"
        R.barrier()
    R.print(f"✅ notify set / add on {w} ranks x{args.rounds} passed")
    R.close()


if __name__ == "__main__":
    main()
