"""notify / wait ring queue between neighbouring ranks (20 workgroups = 16 producers + 4 consumers, 2000 repeats).

    bash scripts/launch_amd.sh python/triton_dist/test/amd/test_distributed-notify-wait.py
    python triton_dist/test/amd/test_distributed-notify-wait.py --virtual-ranks 4 --repeats 50

Rank r pushes items into rank (r+1)'s symmetric queue and pops its own; a slot is handed over with
notify(signal, peer, 2*round+1) / wait(signal == 2*round+1) / notify(signal, self, 2*round+2).
(Scenario of reference python/triton_dist/test/amd/test_distributed-notify-wait.py:39-212.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
os.environ.setdefault("RANDOM_PARAMS", "1")

import torch  # noqa: E402

from triton_distributed_gh_amd.ops import primitives  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queue_size", type=int, default=8)
    ap.add_argument("--item_elems", type=int, default=512)
    ap.add_argument("--items", type=int, default=128)
    ap.add_argument("--repeats", type=int, default=2000)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    assert R.device.type == "cuda", "the ring-queue kernel needs a GPU"
    w, Q, E, N = R.world_size, args.queue_size, args.item_elems, args.items
    queues = R.collective(lambda r, env: env.heap.create_tensor([Q * E], torch.int32))
    sigs = R.collective(lambda r, env: env.heap.create_tensor([Q], torch.int32, arena="sig"))
    errs = {r: torch.zeros(4, dtype=torch.int32, device=R.device) for r in R.local_ranks}
    for rep in range(args.repeats):
        ins = {r: torch.randint(0, 1 << 30, (N * E,), device=R.device, dtype=torch.int32) for r in R.local_ranks}
        outs = {r: torch.zeros(N * E, device=R.device, dtype=torch.int32) for r in R.local_ranks}
        for r in R.local_ranks:
            sigs[r].zero_()
        R.barrier()
        R.run(lambda r, env: primitives.ring_queue(queues[r], sigs[r], ins[r], outs[r], (r + 1) % w, Q, E, N, 16, 4,
                                                   err=errs[r], env=env))
        R.barrier()
        sent = R.all_gather(ins)
        for r in R.local_ranks:
            assert int(errs[r][0]) == 0, f"rank {r}: a wait timed out"
            assert torch.equal(outs[r], sent[(r - 1) % w]), f"repeat {rep}: rank {r} popped wrong data"
    R.print(f"✅ notify/wait ring queue: {args.repeats} repeats x {N} items on {w} ranks passed")
    R.close()


if __name__ == "__main__":
    main()
