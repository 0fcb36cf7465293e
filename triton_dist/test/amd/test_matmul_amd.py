"""The gfx950 MFMA GEMM (C = A @ B^T, bf16 / fp16, fp32 accumulate) vs torch.matmul: correctness
and TFLOP/s next to hipBLASLt.

    python triton_dist/test/amd/test_matmul_amd.py -M 4096 -N 8192 -K 8192
    python triton_dist/test/amd/test_matmul_amd.py --check

(Scenario of reference python/triton_dist/test/amd/test_matmul_amd.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import time  # noqa: E402

from triton_distributed_gh_amd.ops.gemm import gemm_nt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-M", type=int, default=4096)
    ap.add_argument("-N", type=int, default=8192)
    ap.add_argument("-K", type=int, default=8192)
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float16"])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs a GPU"
    dt = getattr(torch, args.dtype)
    shapes = [(args.M, args.N, args.K)] if not args.check else [(256, 256, 128), (520, 776, 1024), (4096, 1024, 512),
                                                              (8, 4096, 4096), (1000, 1000, 320)]
    for M, N, K in shapes:
        a = (torch.rand(M, K, device="cuda") - 0.5).to(dt)
        b = (torch.rand(N, K, device="cuda") - 0.5).to(dt)
        c = gemm_nt(a, b)
        if args.check:
            ref = a.float() @ b.float().t()
            err = (c.float() - ref).abs().max().item() / ref.abs().max().item()
            assert err < 1e-2, f"{M}x{N}x{K}: {err}"
            continue
        out = torch.empty(M, N, device="cuda", dtype=dt)
        for name, fn in (("gfx950 MFMA kernel", lambda: gemm_nt(a, b, out=out)),
                         ("torch.matmul (hipBLASLt)", lambda: torch.matmul(a, b.t(), out=out))):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                fn()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / args.iters * 1e6
            print(f"{name:>26}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
    if args.check:
        print(f"✅ gemm_nt {args.dtype} vs fp32 reference on {len(shapes)} shapes passed", flush=True)


if __name__ == "__main__":
    main()
