"""Tutorial 07 - overlapping all-gather with GEMM (the NVIDIA-named entry points).

C[M, N/w] = AllGather_M(A[M/w, K]) @ B[N/w, K]^T.  A push kernel writes every rank's rows into the
peers' symmetric workspaces chunk by chunk and raises one flag per (source rank, chunk); the
persistent MFMA GEMM pulls tile jobs in expected-arrival order (own rows first) and waits only for the
flags its tile needs - compute starts immediately and the transfer hides behind it
(csrc/ops/ag_gemm.hip).  Calls are numbered by an epoch and buffers alternate, so no reset or barrier
follows the GEMM.

Part 1 (GPU only) is that protocol as a USER kernel - tutorials/kernels/ag_gemm_user.hip, compiled here with hipcc:
    producers :  per 256-row chunk: write-through stores into symm_at(ws, peer) for every peer; drain; barrier;
                 count on done[chunk]; the completing producer notify()s flags[rank * cpr + chunk] on every rank
    tile (m,n):  token = wait(flags + src_rank * cpr + chunk, 1, SYS, ACQUIRE, epoch)
                 TileArgs t{A = consume_token(ws, token) + m * 256 * K, B = ...};
                 mainloop_256_2ph<DT_BF16, true, 3>(t, smem, acc);      // tdist/gemm_gfx950.hpp: LDS-DMA + MFMA
                 epilogue_256_direct<true>(acc, store);
Part 2 is the library op: the same protocol in a persistent kernel with per-XCD work queues, an in-kernel or
copy-engine producer and device-side call numbering.

    python tutorials/07-overlapping-allgather-gemm.py --virtual-ranks 4 --M 2048 --N 2048 --K 1024
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402,F401

from triton_dist.kernels.nvidia import ag_gemm, create_ag_gemm_context  # noqa: E402
from triton_dist.tools import compile_hip, load_module  # noqa: E402


def user_kernel_ag_gemm(R):
    """Part 1: tutorials/kernels/ag_gemm_user.hip - producers, chunk flags and the wait in front of the tile engine's
    main loop, written against the public headers and launched as a code object.  Fixed small shape: one 256-row
    chunk per rank, N = 512, K = 1024, so every workgroup of every rank (2 producers + 2 w tiles) is resident even
    when all ranks share one GPU."""
    w, dev = R.world_size, R.device
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernels", "ag_gemm_user.hip")
    kernel = load_module(compile_hip(src)).get("ag_gemm_user")
    M, N, K, push_wgs = 256 * w, 512, 1024, 2
    tiles = (M // 256) * (N // 256)
    dt = torch.bfloat16
    ws = R.collective(lambda r, env: env.heap.create_tensor([M, K], dt))
    flags = R.collective(lambda r, env: env.heap.create_tensor([w], torch.int32, arena="sig"))
    A = {r: (torch.rand(M // w, K, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    B = {r: (torch.rand(N, K, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    C = {r: torch.zeros(M, N, device=dev, dtype=dt) for r in R.local_ranks}
    done = {r: torch.zeros(8, dtype=torch.int32, device=dev) for r in R.local_ranks}
    err = {r: torch.zeros(4, dtype=torch.int32, device=dev) for r in R.local_ranks}
    R.barrier()
    R.run(lambda r, env: kernel(push_wgs + tiles, 512, torch.cuda.current_stream(), env.heap, A[r], B[r], C[r], ws[r],
                                flags[r], done[r], M, N, K, push_wgs, 1, err[r], lds_bytes=135168))
    full = torch.cat(R.all_gather(A)).float()
    for r in R.local_ranks:
        assert int(err[r][0]) == 0 and rel_err(C[r], full @ B[r].float().t()) < 1e-2
    R.print("user-kernel AG + GEMM ok")


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
        user_kernel_ag_gemm(R)
    dt = torch.bfloat16
    A = {r: (torch.rand(M // w, K, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    B = {r: (torch.rand(N // w, K, device=dev) - 0.5).to(dt) for r in R.local_ranks}
    ctxs = R.collective(lambda r, env: create_ag_gemm_context(A[r], B[r], r, w, max_M=M, env=env))
    R.barrier()
    out = R.run(lambda r, env: ag_gemm(A[r], B[r], ctxs[r], straggler_option=(w - 1, 5_000_000) if dev.type == "cuda" else None))
    full = torch.cat(R.all_gather(A)).float()
    for r in R.local_ranks:
        assert rel_err(out[r], full @ B[r].float().t()) < 1e-2
    t = R.time(lambda r, env: ag_gemm(A[r], B[r], ctxs[r]), args.warmup, args.iters)
    R.print(f"✅ tutorial 07: ag_gemm M={M} N={N} K={K} on {w} ranks correct, {t:.3f} ms per call")
    R.close()


if __name__ == "__main__":
    main()
