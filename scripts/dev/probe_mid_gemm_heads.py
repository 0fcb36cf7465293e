"""LM-head shaped GEMMs (the my_lab/benchmark sweep) at batch 32 / 64 / 128: mid-M kernel, 256x256 tile kernel
and hipBLASLt (torch.matmul), microseconds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from triton_distributed_gh_amd.ops.gemm import GEMM_VARIANT, gemm_nt, gemm_nt_mid  # noqa: E402
from probe_mid_gemm import timeit  # noqa: E402

HEADS = [("mistral-7b", 32768, 4096), ("llama3-8b", 128256, 4096), ("llama3-70b", 12800, 8192),
         ("qwen3-30b", 151936, 2048), ("Gemma-7b", 256000, 3072)]


def main():
    dt = torch.bfloat16
    wins = total = 0
    for name, N, K in HEADS:
        w = (torch.rand(N, K, device="cuda") * 2 - 1).to(dt)
        for M in (32, 64, 128):
            x = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt)
            o = torch.empty(M, N, device="cuda", dtype=dt)
            mid = min(timeit(lambda: gemm_nt_mid(x, w, o), n=15, warm=4) for _ in range(2))
            big = min(timeit(lambda: gemm_nt(x, w, out=o, force_native=True, variant=GEMM_VARIANT | (2 << 8)), n=15, warm=4)
                      for _ in range(1))
            lib = min(timeit(lambda: torch.matmul(x, w.t(), out=o), n=15, warm=4) for _ in range(2))
            wins += mid <= lib
            total += 1
            print(f"{name:12s} N={N:6d} K={K:5d} M={M:3d}: mid {mid:7.1f}  tile256 {big:7.1f}  lib {lib:7.1f}  "
                  f"({lib / mid:4.2f}x, {N * K * 2 / mid / 1e6:4.2f} TB/s)", flush=True)
        del w
    print(f"mid <= library on {wins} of {total}")


if __name__ == "__main__":
    main()
