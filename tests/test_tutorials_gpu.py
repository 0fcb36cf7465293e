"""The tutorial kernels (tutorials/kernels/*_user.hip): stand-alone HIP files that include only the public device
headers, compiled with hipcc at test time, loaded as code objects and launched on 4 virtual ranks.

Scenarios: the reference's tutorials 02 (all-gather), 05 (reduce-scatter), 07 / 09 (AG + GEMM) and 08 / 10
(GEMM + RS) - each of which ends in a comparison against torch (e.g. tutorials/07-overlapping-allgather-gemm.py:442-529)."""
import time
from pathlib import Path

import pytest
import torch

from triton_distributed_gh_amd.tools import runtime as rt
from triton_distributed_gh_amd.utils.testing import VirtualCluster

pytestmark = pytest.mark.gpu
KERNELS = Path(__file__).resolve().parent.parent / "tutorials" / "kernels"
W = 4
LDS = 135168          # tdist::gemm::LDS_BYTES


def _kernel(name):
    return rt.load_module(rt.compile_hip(KERNELS / f"{name}.hip")).get(name)


def test_allgather_push_user_kernel():
    k = _kernel("allgather_push_user")
    n16, slices = 4096 + 3, 4                      # 64 KiB + a ragged tail per shard
    vc = VirtualCluster(W, heap_bytes=1 << 24)
    try:
        gathered = vc.collective(lambda r, env: env.heap.create_tensor([W, n16 * 4], torch.int32))
        flags = vc.collective(lambda r, env: env.heap.create_tensor([W], torch.int32, arena="sig"))
        done = [torch.zeros(W, dtype=torch.int32, device="cuda") for _ in range(W)]
        errs = [torch.zeros(4, dtype=torch.int32, device="cuda") for _ in range(W)]
        for epoch in (1, 2, 3):
            shards = [torch.randint(-2 ** 31, 2 ** 31 - 1, (n16 * 4,), dtype=torch.int32, device="cuda") for _ in range(W)]
            torch.cuda.synchronize()
            vc.run(lambda r, env: k(W * slices, 256, torch.cuda.current_stream(), env.heap, shards[r], gathered[r],
                                    flags[r], done[r], n16, slices, epoch, errs[r]))
            for r in range(W):
                assert int(errs[r][0]) == 0, "a wait timed out"
                assert torch.equal(gathered[r], torch.stack(shards)), f"epoch {epoch}, rank {r}"
    finally:
        vc.close()


def test_reduce_scatter_user_kernel():
    k = _kernel("reduce_scatter_user")
    n = 16384
    vc = VirtualCluster(W, heap_bytes=1 << 24)
    try:
        scatter = vc.collective(lambda r, env: env.heap.create_tensor([W, n], torch.float32))
        flags = vc.collective(lambda r, env: env.heap.create_tensor([W], torch.int32, arena="sig"))
        errs = [torch.zeros(4, dtype=torch.int32, device="cuda") for _ in range(W)]
        outs = [torch.zeros(n, device="cuda") for _ in range(W)]
        for epoch in (1, 2):
            parts = [torch.randint(-1000, 1000, (W, n), device="cuda").float() for _ in range(W)]    # exact in fp32
            torch.cuda.synchronize()
            vc.run(lambda r, env: k(W + 8, 256, torch.cuda.current_stream(), env.heap, parts[r], scatter[r], flags[r],
                                    outs[r], n, epoch, errs[r]))
            total = torch.stack(parts).sum(0)                  # [owner, n]
            for r in range(W):
                assert int(errs[r][0]) == 0
                assert torch.equal(outs[r], total[r]), f"epoch {epoch}, rank {r}"
    finally:
        vc.close()


def test_ag_gemm_user_kernel():
    k = _kernel("ag_gemm_user")
    M, N, K, push_wgs = 2048, 1024, 1024, 4         # 2 chunks per rank; 8 x 4 tiles + 4 producers = 36 workgroups per rank
    tiles = (M // 256) * (N // 256)
    assert W * (tiles + push_wgs) <= 256, "every workgroup of every rank must be resident"
    vc = VirtualCluster(W, heap_bytes=1 << 26)
    try:
        ws = vc.collective(lambda r, env: env.heap.create_tensor([M, K], torch.bfloat16))
        flags = vc.collective(lambda r, env: env.heap.create_tensor([W * (M // W // 256)], torch.int32, arena="sig"))
        done = [torch.zeros(16, dtype=torch.int32, device="cuda") for _ in range(W)]
        errs = [torch.zeros(4, dtype=torch.int32, device="cuda") for _ in range(W)]
        B = [(torch.rand(N, K, device="cuda") - 0.5).bfloat16() for _ in range(W)]
        C = [torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(W)]
        t0 = time.perf_counter()
        for epoch in (1, 2):
            A = [(torch.rand(M // W, K, device="cuda") - 0.5).bfloat16() for _ in range(W)]
            torch.cuda.synchronize()
            vc.run(lambda r, env: k(push_wgs + tiles, 512, torch.cuda.current_stream(), env.heap, A[r], B[r], C[r], ws[r],
                                    flags[r], done[r], M, N, K, push_wgs, epoch, errs[r], lds_bytes=LDS))
            full = torch.cat(A).float()
            for r in range(W):
                assert int(errs[r][0]) == 0
                ref = full @ B[r].float().t()
                assert (C[r].float() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item() + 1e-2, f"rank {r}"
        assert time.perf_counter() - t0 < 10
    finally:
        vc.close()


def test_gemm_rs_user_kernel():
    k = _kernel("gemm_rs_user")
    M, N, K = 2048, 1024, 512                       # K is the per-rank slice
    tiles, R = (M // 256) * (N // 256), 8
    assert W * (tiles + R) <= 256
    vc = VirtualCluster(W, heap_bytes=1 << 26)
    try:
        scatter = vc.collective(lambda r, env: env.heap.create_tensor([W, M // W, N], torch.bfloat16))
        flags = vc.collective(lambda r, env: env.heap.create_tensor([W], torch.int32, arena="sig"))
        done = [torch.zeros(W, dtype=torch.int32, device="cuda") for _ in range(W)]
        errs = [torch.zeros(4, dtype=torch.int32, device="cuda") for _ in range(W)]
        outs = [torch.zeros(M // W, N, device="cuda", dtype=torch.bfloat16) for _ in range(W)]
        for epoch in (1, 2):
            A = [(torch.rand(M, K, device="cuda") - 0.5).bfloat16() for _ in range(W)]
            B = [(torch.rand(N, K, device="cuda") - 0.5).bfloat16() for _ in range(W)]
            torch.cuda.synchronize()
            vc.run(lambda r, env: k(tiles + R, 512, torch.cuda.current_stream(), env.heap, A[r], B[r], scatter[r], flags[r],
                                    done[r], outs[r], M, N, K, tiles, epoch, errs[r], lds_bytes=LDS))
            # the kernel rounds every rank's partial to bf16 before the fp32 sum
            ref = sum((A[s].float() @ B[s].float().t()).bfloat16().float() for s in range(W))
            for r in range(W):
                assert int(errs[r][0]) == 0
                got, want = outs[r].float(), ref[r * (M // W):(r + 1) * (M // W)]
                assert (got - want).abs().max().item() <= 2.0 ** -7 * want.abs().max().item() + 1e-2, f"rank {r}"
    finally:
        vc.close()
