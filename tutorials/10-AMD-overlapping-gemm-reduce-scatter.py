"""Tutorial 10 - AMD entry points: create_gemm_rs_intra_node_context / gemm_rs_intra_node.

out[M/w, N] = ReduceScatter_M(A[M, K/w] @ B[N, K/w]^T).  ONE persistent kernel: tiles are produced in
destination order (the peer "after" this rank first, own segment last), the epilogue stores each tile
straight into the owning rank's symmetric buffer over xGMI, per-chunk counters raise an epoch flag on
the destination, and workgroups that run out of GEMM tiles reduce chunks whose w partials have
arrived - the reduction of early chunks overlaps the GEMM of late ones (csrc/ops/gemm_rs.hip).

Part 1 (GPU only) is that protocol as a USER kernel - tutorials/kernels/gemm_rs_user.hip, compiled here with hipcc:
    tile (m,n):  mainloop_256_2ph<DT_BF16, true, 3>(t, smem, acc);
                 epilogue_256_direct<true>(acc, [&](r, c, v) { st_wt128_sys(symm_at(scatter, owner)[rank] + ..., v); });
                 drain_stores(); barrier; count on done[owner]; the completing tile notify()s flags[rank] on owner
    reducers  :  token = wait(flags, w, SYS, ACQUIRE, epoch); out = sum over the w slots (bf16 partials, fp32 sum)
Part 2 is the library op: persistent, reduction of early chunks overlapped with the GEMM of late ones.

    python tutorials/10-AMD-overlapping-gemm-reduce-scatter.py --virtual-ranks 4 --M 2048 --N 1024 --K 2048
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402,F401

from triton_dist.kernels.amd import create_gemm_rs_intra_node_context, gemm_rs_intra_node  # noqa: E402
from triton_dist.tools import compile_hip, load_module  # noqa: E402


def user_kernel_gemm_rs(R):
    """Part 1: tutorials/kernels/gemm_rs_user.hip - the GEMM whose epilogue stores every tile into the owner's
    scatter slot over xGMI, the per-owner counter + notify, and the workgroups that wait for all w partials and sum
    them.  Fixed small shape: 256 rows per rank, N = 512, K/w = 512."""
    w, dev = R.world_size, R.device
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernels", "gemm_rs_user.hip")
    kernel = load_module(compile_hip(src)).get("gemm_rs_user")
    M, N, K, reducers = 256 * w, 512, 512, 4
    tiles = (M // 256) * (N // 256)
    dt = torch.bfloat16
    scatter = R.collective(lambda r, env: env.heap.create_tensor([w, M // w, N], dt))
    flags = R.collective(lambda r, env: env.heap.create_tensor([w], torch.int32, arena="sig"))
    A = {r: (torch.rand(M, K, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    B = {r: (torch.rand(N, K, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    out = {r: torch.zeros(M // w, N, device=dev, dtype=dt) for r in R.local_ranks}
    done = {r: torch.zeros(w, dtype=torch.int32, device=dev) for r in R.local_ranks}
    err = {r: torch.zeros(4, dtype=torch.int32, device=dev) for r in R.local_ranks}
    R.barrier()
    R.run(lambda r, env: kernel(tiles + reducers, 512, torch.cuda.current_stream(), env.heap, A[r], B[r], scatter[r],
                                flags[r], done[r], out[r], M, N, K, tiles, 1, err[r], lds_bytes=135168))
    total = sum(p.float() for p in R.all_gather({r: (A[r].float() @ B[r].float().t()).to(dt) for r in R.local_ranks}))
    for r in R.local_ranks:
        assert int(err[r][0]) == 0 and rel_err(out[r], total[r * (M // w):(r + 1) * (M // w)]) < 1e-2
    R.print("user-kernel GEMM + reduce-scatter ok")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--N", type=int, default=8192)
    ap.add_argument("--K", type=int, default=8192)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, M, N, K = R.world_size, R.device, args.M, args.N, args.K
    if dev.type == "cuda":
        user_kernel_gemm_rs(R)
    dt = torch.bfloat16
    A = {r: (torch.rand(M, K // w, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    B = {r: (torch.rand(N, K // w, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    ctxs = R.collective(lambda r, env: create_gemm_rs_intra_node_context(M, N, dt, r, w, env.tp_group, env=env))
    R.barrier()
    out = R.run(lambda r, env: gemm_rs_intra_node(A[r], B[r], ctxs[r]))
    total = sum(p.float() for p in R.all_gather({r: (A[r].float() @ B[r].float().t()).to(dt) for r in R.local_ranks}))
    for r in R.local_ranks:
        assert rel_err(out[r], total[r * (M // w):(r + 1) * (M // w)]) < 1e-2
    t = R.time(lambda r, env: gemm_rs_intra_node(A[r], B[r], ctxs[r]), args.warmup, args.iters)
    R.print(f"✅ tutorial 10: gemm_rs_intra_node M={M} N={N} K={K} on {w} ranks correct, {t:.3f} ms per call")
    R.close()


if __name__ == "__main__":
    main()
