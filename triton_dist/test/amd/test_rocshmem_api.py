"""Symmetric-heap device API smoke test: putmem_signal_block to every peer, signal_wait_until,
getmem_block back - the device functions the reference reaches through rocSHMEM.

    bash scripts/launch_amd.sh python/triton_dist/test/amd/test_rocshmem_api.py
    python triton_dist/test/amd/test_rocshmem_api.py --virtual-ranks 4

(Scenario of reference python/triton_dist/test/amd/test_rocshmem_api.py; host API: pyrocshmem.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import pyrocshmem  # noqa: E402
from triton_distributed_gh_amd.ops import primitives  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=4096, help="ints per put")
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    assert R.device.type == "cuda", "needs a GPU"
    w, n = R.world_size, args.n
    bufs = R.collective(lambda r, env: env.heap.create_tensor([w * n], torch.int32))
    sigs = R.collective(lambda r, env: env.heap.create_tensor([w], torch.int64, arena="sig"))
    srcs = {r: torch.full((n,), 100 + r, dtype=torch.int32, device=R.device) for r in R.local_ranks}
    backs = {r: torch.zeros(w * n, dtype=torch.int32, device=R.device) for r in R.local_ranks}
    R.barrier()
    R.run(lambda r, env: primitives.shmem_api_smoke(bufs[r], sigs[r], srcs[r], backs[r], n, 77, env=env))
    R.barrier()
    for r in R.local_ranks:
        assert bufs[r].view(w, n)[:, 0].tolist() == [100 + s for s in range(w)], "slot s must hold rank s's value"
        assert sigs[r].tolist() == [77] * w
        assert backs[r].view(w, n)[:, 0].tolist() == [100 + r] * w, "getmem must read our slot back from every peer"
    R.print(f"✅ putmem_signal / signal_wait_until / getmem on {w} ranks passed "
            f"(pyrocshmem host API: {len([x for x in dir(pyrocshmem) if x.startswith('rocshmem')])} entry points)")
    R.close()


if __name__ == "__main__":
    main()
