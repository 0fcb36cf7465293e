"""Tutorial 01 - notify / wait: a producer-consumer ring queue between neighbouring ranks.

The primitives (reference: docs/primitives.md, tutorials/01-distributed-notify-wait.py:62-145):

    wait(flags, n, scope, semantic, value)   workgroup-collective spin until n flags == value
    consume_token(x, token)                  ties later uses of x to that wait
    notify(ptr, rank, signal, op, scope)     make earlier stores visible, then set / add the signal on rank
    symm_at(ptr, rank)                       the same symmetric object on another rank

In device code they are HIP functions (csrc/include/tdist/primitives.hpp):

    // producer workgroup, item i -> slot i % Q, round r = i / Q
    wait(symm_at(ctx, signal, peer) + slot, 1, SCOPE_SYS, SEM_ACQUIRE, 2 * r);        // slot free?
    copy item into symm_at(ctx, queue, peer) + slot * elems;
    fence(release, system); s_waitcnt vmcnt(0); __syncthreads();
    if (threadIdx.x == 0) notify(ctx, signal + slot, peer, 2 * r + 1, SIGNAL_SET, COMM_INTRA_NODE);
    // consumer workgroup
    wait(signal + slot, 1, SCOPE_SYS, SEM_ACQUIRE, 2 * r + 1);                        // slot full?
    read queue + slot * elems;  __syncthreads();
    if (threadIdx.x == 0) notify(ctx, signal + slot, rank, 2 * r + 2, ...);           // hand it back

Part 1 writes that kernel the way a USER of the framework does: tutorials/kernels/ring_queue_user.hip is
a stand-alone HIP file that includes nothing but tdist/primitives.hpp; it is compiled here with hipcc
(triton_dist.tools.compile_hip), loaded as a code object (load_module) and launched from Python with the
symmetric heap as its SymmCtx argument.  Part 2 replays the same protocol from the HOST with
triton_dist.language (stream-ordered on a GPU, immediate on the CPU backend) - handy for prototyping a
protocol before writing the kernel.

    python tutorials/01-distributed-notify-wait.py --virtual-ranks 4
    bash scripts/launch_amd.sh tutorials/01-distributed-notify-wait.py
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402,F401

import triton_dist.language as dl  # noqa: E402
from triton_dist.tools import compile_hip, load_module  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev = R.world_size, R.device
    Q, E, N = 4, 256, 32
    queues = R.collective(lambda r, env: env.heap.create_tensor([Q * E], torch.int32))
    sigs = R.collective(lambda r, env: env.heap.create_tensor([Q], torch.int32, arena="sig"))
    ins = {r: torch.arange(N * E, dtype=torch.int32, device=dev) + 1000 * r for r in R.local_ranks}
    outs = {r: torch.zeros(N * E, dtype=torch.int32, device=dev) for r in R.local_ranks}
    if dev.type == "cuda":          # ---- part 1: the user's device kernel, built and loaded at run time
        src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernels", "ring_queue_user.hip")
        kernel = load_module(compile_hip(src)).get("ring_queue_user")
        errs = {r: torch.zeros(4, dtype=torch.int32, device=dev) for r in R.local_ranks}
        R.barrier()
        # grid = 4 producer + 2 consumer workgroups; arguments in the kernel's parameter order
        R.run(lambda r, env: kernel(6, 256, torch.cuda.current_stream(), env.heap, queues[r], sigs[r], ins[r], outs[r],
                                    (r + 1) % w, Q, E, N, 0, 4, 2, errs[r]))
        sent = R.all_gather(ins)
        for r in R.local_ranks:
            assert int(errs[r][0]) == 0, "a wait timed out"
            assert torch.equal(outs[r], sent[(r - 1) % w])
        R.print("user kernel ring queue: every rank received its left neighbour's items")
    # ---- part 2: the same hand-over driven from the host, one item
    for r in R.local_ranks:
        sigs[r].zero_()
    R.barrier()

    def host_protocol(r, env):
        peer = (r + 1) % w
        dl.symm_at(queues[r], peer, env=env)[:E].copy_(ins[r][:E])                   # store into the peer's slot 0
        dl.notify(sigs[r], peer, signal=1, sig_op="set", comm_scope="intra_node", env=env)
        token = dl.wait(sigs[r], 1, "sys", "acquire", waitValue=1, env=env)          # our own slot 0 is full
        return dl.consume_token(queues[r], token)[:E].clone()

    got = R.run(host_protocol)
    sent = R.all_gather(ins)
    for r in R.local_ranks:
        assert torch.equal(got[r], sent[(r - 1) % w][:E])
    R.print("✅ host-level notify / wait hand-over passed")
    R.close()


if __name__ == "__main__":
    main()
