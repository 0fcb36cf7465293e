"""Tutorial 05 - intra-node reduce-scatter in one hop.

Every rank publishes its [M, N] partial in a symmetric buffer and flags its peers; each rank then
reads ITS row segment from all w buffers over w-1 different xGMI links at once and adds in fp32
(csrc/ops/reduce_scatter.hip).  The reference's tutorial 05 scatters with the copy engine and then
ring-reduces; on a full mesh the ring only adds hops.

Part 1 (GPU only) is the scatter + reduce protocol as a USER kernel, tutorials/kernels/reduce_scatter_user.hip
(fp32, 40 lines against tdist/primitives.hpp, compiled here with hipcc):
    workgroup d < w :  copy partial[d] into symm_at(scatter, d)[rank]; drain; barrier; notify(flags + rank, d, epoch)
    the other R     :  token = wait(flags, w, SYS, ACQUIRE, epoch); out = sum over consume_token(scatter, token)[s]
The pushing workgroups never wait, so a waiting one can only be waiting for another rank.  Part 2 is the library
op (bf16, one-hop pull form).

    python tutorials/05-intra-node-reduce-scatter.py --virtual-ranks 4
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402,F401

from triton_dist.kernels.nvidia.reduce_scatter import create_reduce_scater_2d_ctx, reduce_scatter_2d_op  # noqa: E402
from triton_dist.tools import compile_hip, load_module  # noqa: E402


def user_kernel_reduce_scatter(R, n=1 << 14):
    w, dev = R.world_size, R.device
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernels", "reduce_scatter_user.hip")
    kernel = load_module(compile_hip(src)).get("reduce_scatter_user")
    scatter = R.collective(lambda r, env: env.heap.create_tensor([w, n], torch.float32))
    flags = R.collective(lambda r, env: env.heap.create_tensor([w], torch.int32, arena="sig"))
    parts = {r: torch.randint(-1000, 1000, (w, n), device=dev).float() for r in R.local_ranks}     # exact in fp32
    outs = {r: torch.zeros(n, device=dev) for r in R.local_ranks}
    err = {r: torch.zeros(4, dtype=torch.int32, device=dev) for r in R.local_ranks}
    R.barrier()
    R.run(lambda r, env: kernel(w + 8, 256, torch.cuda.current_stream(), env.heap, parts[r], scatter[r], flags[r], outs[r],
                                n, 1, err[r]))
    total = sum(R.all_gather(parts))                    # [owner, n]
    for r in R.local_ranks:
        assert int(err[r][0]) == 0 and torch.equal(outs[r], total[r])
    R.print("user-kernel reduce-scatter ok (bit-exact)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=2048)
    ap.add_argument("--N", type=int, default=1024)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, M, N = R.world_size, R.device, args.M, args.N
    dt = torch.bfloat16 if dev.type == "cuda" else torch.float32
    if dev.type == "cuda":
        user_kernel_reduce_scatter(R)
    ctxs = R.collective(lambda r, env: create_reduce_scater_2d_ctx(M, N, r, w, w, dt, env=env))
    x = {r: (torch.rand(M, N, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    R.barrier()
    out = R.run(lambda r, env: reduce_scatter_2d_op(x[r], ctxs[r]))
    total = sum(t.float() for t in R.all_gather(x))
    for r in R.local_ranks:
        assert rel_err(out[r], total[r * (M // w):(r + 1) * (M // w)]) < 2e-2
    R.print(f"✅ tutorial 05: reduce-scatter of [{M}, {N}] over {w} ranks matches the fp32 sum")
    R.close()


if __name__ == "__main__":
    main()
