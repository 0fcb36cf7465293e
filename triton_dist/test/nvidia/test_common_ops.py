"""Barriers and stream signals: barrier_all_on_stream across ranks (with a straggler), the device
grid barrier, and set_signal / wait_eq stream memory operations.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_common_ops.py
    python triton_dist/test/nvidia/test_common_ops.py --virtual-ranks 4

(CLI of reference python/triton_dist/test/nvidia/test_common_ops.py and test/amd/test_barrier_all.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.nvidia.common_ops import (BarrierAllContext, barrier_all_on_stream,  # noqa: E402
                                                   barrier_on_this_grid_smoke, set_signal, wait_eq)
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=50)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev = R.world_size, R.device
    ctxs = R.collective(lambda r, env: BarrierAllContext(env=env))
    board = R.collective(lambda r, env: env.heap.create_tensor([w], torch.int32))
    for it in range(args.rounds):
        def step(r, env):
            if dev.type == "cuda" and it % 7 == r % 7:
                torch.cuda._sleep(2_000_000)                  # straggler
            for p, view in enumerate(env.heap.peer_views(board[r])):
                view[r] = it + 1                               # write my round number on every rank's board
            barrier_all_on_stream(ctxs[r])
            return board[r].clone()

        seen = R.run(step)
        for r in R.local_ranks:
            assert seen[r].tolist() == [it + 1] * w, f"round {it} rank {r}: {seen[r].tolist()}"
    R.print(f"✅ barrier_all_on_stream x{args.rounds} world={w} passed")
    if dev.type == "cuda":
        for blocks in (8, 256, 1024):
            assert barrier_on_this_grid_smoke(blocks, rounds=16) == (0, 0)
        R.print("✅ barrier_on_this_grid (8 / 256 / 1024 workgroups) passed")
        sig = torch.zeros(1, dtype=torch.int32, device=dev)
        set_signal(sig, 5)
        wait_eq(sig, 5)
        torch.cuda.synchronize()
        assert int(sig) == 5
        R.print("✅ set_signal / wait_eq stream operations passed")
    R.close()


if __name__ == "__main__":
    main()
