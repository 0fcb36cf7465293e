"""Tutorial 02 - intra-node all-gather three ways on a full-mesh xGMI node.

One MI355X node connects every GPU to every other GPU directly (7 links per GPU), so the right
all-gather is ONE hop: every rank pushes (or pulls) w-1 shards over w-1 different links at once; a
ring would use one link per direction.  (Reference tutorial 02 and kernels/nvidia/allgather.py use
copy-engine push / pull and ring variants for NVLink topologies.)

  0. the protocol itself as a USER kernel (GPU only): tutorials/kernels/allgather_push_user.hip - 40 lines against
     tdist/primitives.hpp + tdist/cache_policy.hpp, compiled here with hipcc and launched through load_module:
         workgroup (destination d, slice s):  write-through stores of slice s into symm_at(gathered, d)[rank]
                                              drain_stores(); __syncthreads(); count the slice on done[d]
         last slice of d:                     notify(flags + rank, d, epoch)          // flag ON rank d
         slice 0 of d:                        wait(flags + d, 1, SYS, ACQUIRE, epoch) // rank d's shard is HERE
     Why write-through stores: the slices of one destination run on different XCDs whose L2s are not coherent, and
     the release fence inside notify() only writes back the notifying workgroup's own L2.
  1. copy engines: cp_engine_producer_all_gather_intra_node - SDMA copies + stream-ordered flags
  2. push kernel : fast_allgather(mode="push_2d")           - peer stores + epoch flag
  3. LL protocol : fast_allgather(mode="push_2d_ll")        - every 4 payload bytes travel with a flag

    python tutorials/02-intra-node-allgather.py --virtual-ranks 4
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402,F401

from triton_dist.kernels.nvidia import (cp_engine_producer_all_gather_intra_node, create_fast_allgather_context,  # noqa: E402
                                        fast_allgather)
from triton_dist.tools import compile_hip, load_module  # noqa: E402

KERNELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernels")


def user_kernel_allgather(R, shard, want, m, n):
    """Part 0: the push all-gather written as a user kernel."""
    w, dev = R.world_size, R.device
    kernel = load_module(compile_hip(os.path.join(KERNELS, "allgather_push_user.hip"))).get("allgather_push_user")
    n16, slices = m * n * 2 // 16, 4
    gathered = R.collective(lambda r, env: env.heap.create_tensor([w, m * n], torch.bfloat16))
    flags = R.collective(lambda r, env: env.heap.create_tensor([w], torch.int32, arena="sig"))
    done = {r: torch.zeros(w, dtype=torch.int32, device=dev) for r in R.local_ranks}
    err = {r: torch.zeros(4, dtype=torch.int32, device=dev) for r in R.local_ranks}
    R.barrier()
    for epoch in (1, 2):            # a new epoch per call: flags are never reset
        R.run(lambda r, env: kernel(w * slices, 256, torch.cuda.current_stream(), env.heap, shard[r], gathered[r],
                                    flags[r], done[r], n16, slices, epoch, err[r]))
    for r in R.local_ranks:
        assert int(err[r][0]) == 0 and torch.equal(gathered[r].view(w * m, n).cpu(), want)
    R.print("user-kernel all-gather ok")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=256)
    ap.add_argument("--cols", type=int, default=1024)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev = R.world_size, R.device
    m, n = args.rows, args.cols
    shard = {r: torch.full((m, n), float(r + 1), device=dev).bfloat16() for r in R.local_ranks}
    want = torch.cat([torch.full((m, n), float(r + 1)) for r in range(w)]).bfloat16()
    bufs = R.collective(lambda r, env: env.heap.create_tensor_list([w * m, n], torch.bfloat16))
    flags = R.collective(lambda r, env: env.heap.create_tensor_list([w], torch.int32, arena="sig"))
    if dev.type == "cuda" and (m * n * 2) % 16 == 0:
        user_kernel_allgather(R, shard, want, m, n)
    # 1. copy engines (flags become 1 once the data of that source rank landed)
    R.barrier()
    R.run(lambda r, env: cp_engine_producer_all_gather_intra_node(r, w, shard[r], bufs[r], flags[r], env=env))
    R.barrier()
    for r in R.local_ranks:
        assert flags[r][r].tolist() == [1] * w and torch.equal(bufs[r][r].cpu(), want)
    R.print("copy-engine all-gather ok")
    # 2 / 3. kernels
    ctxs = R.collective(lambda r, env: create_fast_allgather_context(r, 0, w, 1, max_buffer_size=m * n * 2, env=env))
    sym = R.collective(lambda r, env: env.heap.create_tensor([w, m * n], torch.bfloat16))
    for mode in ("pull", "push_2d", "push_2d_ll"):
        for r in R.local_ranks:
            sym[r].zero_()
            sym[r][r].copy_(shard[r].view(-1))
        R.barrier()
        R.run(lambda r, env: fast_allgather(sym[r], ctxs[r], mode=mode))
        for r in R.local_ranks:
            assert torch.equal(sym[r].view(w * m, n).cpu(), want), mode
        R.print(f"fast_allgather mode={mode} ok")
    R.print("✅ tutorial 02 passed")
    R.close()


if __name__ == "__main__":
    main()
