"""Device-side wait on a symmetric flag released by a peer's notify (set and add flavours), plus
the stream-ordered host API (set_signal / wait_eq).

    bash scripts/launch_amd.sh python/triton_dist/test/amd/test_wait_and_notify.py
    python triton_dist/test/amd/test_wait_and_notify.py --virtual-ranks 2

Rank r's wait kernel blocks its stream until rank r-1's notify lands in r's signal arena.
(Scenario of reference python/triton_dist/test/amd/test_wait_and_notify.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import triton_dist.language as dl  # noqa: E402
from triton_dist.kernels.amd.common_ops import set_signal, wait_eq  # noqa: E402
from triton_distributed_gh_amd.ops import primitives  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    assert R.device.type == "cuda", "the wait kernel needs a GPU"
    w = R.world_size
    flags = R.collective(lambda r, env: env.heap.create_tensor([4], torch.int32, arena="sig"))
    for it in range(1, args.rounds + 1):
        status = {}

        def step(r, env):
            # notify my right neighbour, then wait for my left neighbour's notify
            if it % 2 == 0:
                torch.cuda._sleep(1_000_000 * (r + 1))
            dl.notify(flags[r], (r + 1) % w, signal=it, sig_op="set", comm_scope="intra_node", env=env)
            status[r] = primitives.wait_flag(flags[r], it)

        R.run(step)
        for r in R.local_ranks:
            assert int(status[r][0]) == 1, f"round {it} rank {r}: wait timed out"
            assert int(flags[r][0]) == it
    R.print(f"✅ notify(set) -> wait ring on {w} ranks x{args.rounds} passed")
    # add flavour: everyone adds 1 to rank 0's counter word
    R.barrier()
    R.run(lambda r, env: dl.notify(flags[r][1:], 0, signal=1, sig_op="add", comm_scope="intra_node", env=env))
    R.barrier()
    if 0 in R.local_ranks:
        assert int(flags[0][1]) == w, f"expected {w} adds, saw {int(flags[0][1])}"
    R.print("✅ notify(add) passed")
    sig = torch.zeros(1, dtype=torch.int32, device=R.device)
    out = torch.zeros(1, device=R.device)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s2):
        wait_eq(sig, 9, s2)
        out.fill_(1.0)
    with torch.cuda.stream(s1):
        set_signal(sig, 9, s1)
    torch.cuda.synchronize()
    assert float(out) == 1.0
    R.print("✅ stream-ordered set_signal / wait_eq passed")
    R.close()


if __name__ == "__main__":
    main()
