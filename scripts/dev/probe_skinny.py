"""Decode GEMM (M <= 16): time per shape; run with TDIST_SKINNY_BN=16|32|64 to compare block sizes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from triton_distributed_gh_amd.ops.gemm import gemm_nt

def bench(fn, iters=100, warmup=20):
    for _ in range(warmup): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

shapes = [tuple(int(v) for v in sp.split("x")) for sp in sys.argv[1:]] or [(4096, 4096), (6144, 4096), (4096, 12288), (12800, 8192), (24576, 4096), (32768, 4096), (151936, 2048)]
for M in (1, 16):
    for N, K in shapes:
        a = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
        b = (torch.rand(N, K, device="cuda") - 0.5).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t = bench(lambda: gemm_nt(a, b, out=out, force_native=True))
        tt = bench(lambda: torch.matmul(a, b.t(), out=out))
        print(f"bn={os.environ.get('TDIST_SKINNY_BN', 'auto'):>4} M={M:2d} N={N:6d} K={K:5d}: native {t:7.1f} us ({N * K * 2 / t / 1e6:5.2f} TB/s)  torch {tt:7.1f} us", flush=True)
