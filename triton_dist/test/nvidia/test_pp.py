"""Pipeline-parallel point-to-point primitive (CommOp): rank i streams activations to rank i+1
through double-buffered symmetric buffers with a signal handshake; bit-exact check and bandwidth.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_pp.py --tokens 4096 --hidden 8192
    python triton_dist/test/nvidia/test_pp.py --virtual-ranks 2 --check

(CLI of reference python/triton_dist/test/nvidia/test_pp.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
os.environ.setdefault("RANDOM_PARAMS", "1")

import torch  # noqa: E402

import time  # noqa: E402

from triton_dist.layers.nvidia.p2p import CommOp  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=1024)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--num-buffers", type=int, default=2)
    ap.add_argument("--messages", type=int, default=8)
    ap.add_argument("--sm", type=int, default=8, help="workgroups of the pull kernel")
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, dt = R.world_size, R.device, torch.bfloat16
    assert w >= 2, "needs at least two pipeline stages"
    T, H, NB = args.tokens, args.hidden, args.num_buffers
    ops = R.collective(lambda r, env: CommOp(T, H, r, w, None, dt, num_buffers=NB, env=env))
    g = torch.Generator(device="cpu").manual_seed(args.seed)
    msgs = [torch.randn(T, H, generator=g).to(dev).to(dt) for _ in range(args.messages)]      # same on every rank
    recv = {r: [torch.empty(T, H, device=dev, dtype=dt) for _ in msgs] for r in R.local_ranks}

    def stage(r, env):
        """Stage r forwards what it received (stage 0: the messages) to stage r + 1."""
        for i in range(len(msgs)):
            buf = i % NB
            if r > 0:                                         # receive from r - 1
                ops[r].wait_signal(r - 1, buf, 1)
                ops[r].read(r - 1, buf, recv[r][i], sm=args.sm)
                ops[r].set_signal(r - 1, buf, 0)              # hand the slot back
            if r < w - 1:                                     # send to r + 1 through OUR buffer
                ops[r].wait_signal(r, buf, 0)
                ops[r].get_buffer(buf).copy_(msgs[i] if r == 0 else recv[r][i])
                ops[r].set_signal(r, buf, 1)

    R.barrier()
    if args.check:
        R.run(stage)
        for r in R.local_ranks:
            if r > 0:
                for m, got in zip(msgs, recv[r]):
                    assert torch.equal(m, got), f"stage {r}"
        R.print(f"✅ pp send/recv tokens={T} hidden={H} buffers={NB} stages={w} check passed")
    else:
        R.run(stage)
        R.barrier()
        t0 = time.perf_counter()
        R.run(stage)
        R.barrier()
        dt_s = time.perf_counter() - t0
        nbytes = len(msgs) * T * H * 2
        R.print_all(f"pp pipeline of {len(msgs)} messages x {T * H * 2 / 1e6:.1f} MB over {w} stages: "
                    f"{dt_s * 1e3:.3f} ms ({nbytes / dt_s / 1e9:.1f} GB/s per hop)")
    R.close()


if __name__ == "__main__":
    main()
