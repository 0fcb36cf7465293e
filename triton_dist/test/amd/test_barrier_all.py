"""barrier_all_ipc with staggered arrival (reference: python/triton_dist/test/amd/test_barrier_all.py:142-143
sleeps LOCAL_RANK seconds before the barrier; here every rank delays its stream by a different amount)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
os.environ.setdefault("RANDOM_PARAMS", "1")

import torch  # noqa: E402

from triton_dist.kernels.amd.common_ops import BarrierAllContext, barrier_all_on_stream  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w = R.world_size
    ctxs = R.collective(lambda r, env: BarrierAllContext(env=env))
    marks = R.collective(lambda r, env: env.heap.create_tensor([w], torch.int32))
    for it in range(args.rounds):
        def fn(r, env):
            if env.backend == "hip":
                torch.cuda._sleep(3_000_000 * ((r + it) % w))
            for p in range(w):                                   # publish "rank r reached round it" on every peer
                env.heap.peer_view(marks[r], p)[r] = it + 1
            barrier_all_on_stream(ctxs[r])
            return marks[r].clone()

        out = R.run(fn)
        for r in R.local_ranks:
            assert out[r].tolist() == [it + 1] * w, f"rank {r} left the barrier before everyone arrived: {out[r].tolist()}"
    R.print(f"✅ barrier_all: {args.rounds} staggered rounds on {w} ranks passed")
    R.close()


if __name__ == "__main__":
    main()
