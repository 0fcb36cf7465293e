"""Decode GEMM (M <= 16): the weight-streaming kernel without split-K against the mid-M kernel (split-K across
workgroups, rows padded to 32) and the library, on long-K / short-N layers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from triton_distributed_gh_amd.ops.gemm import gemm_nt, gemm_nt_mid  # noqa: E402
from probe_mid_gemm import timeit  # noqa: E402

SHAPES = [(4096, 12288), (4096, 4096), (5120, 3200), (8192, 3584), (2048, 8192), (1024, 16384), (6144, 4096),
          (24576, 4096), (151936, 2048), (4096, 28672)]
dt = torch.bfloat16
for N, K in SHAPES:
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(dt)
    for M in (1, 8, 16):
        x = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt)
        o = torch.empty(M, N, device="cuda", dtype=dt)
        os.environ["TDIST_DECODE_SPLITK"] = "0"
        sk = min(timeit(lambda: gemm_nt(x, w, out=o, variant=999)) for _ in range(2)) if False else None
        a = min(timeit(lambda: gemm_nt(x, w, out=o)) for _ in range(2))
        b = min(timeit(lambda: gemm_nt_mid(x, w, o)) for _ in range(2))
        c = min(timeit(lambda: torch.matmul(x, w.t(), out=o)) for _ in range(2))
        print(f"N={N:6d} K={K:5d} M={M:2d}: skinny {a:6.1f}  mid {b:6.1f}  lib {c:6.1f}  ({N * K * 2 / min(a, b) / 1e6:4.2f} TB/s best native)",
              flush=True)
    del w
