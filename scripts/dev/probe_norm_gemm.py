"""Fused RMSNorm + decode GEMM, event-timed over a rotation of weight copies larger than the memory-side cache."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.ops.gemm import gemm_nt_norm, gemm_nt

def bench(fn, n=60):
    for _ in range(6): fn(0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

def main():
    dev, bf = "cuda", torch.bfloat16
    for (N, K, silu) in [(6144, 4096, False), (12288, 4096, True)]:
        rows = 2 * N if silu else N
        copies = max(2, int(400e6 / (rows * K * 2)) + 1)
        ws = [(torch.randn(rows, K, device=dev) * 0.02).to(bf) for _ in range(copies)]
        g = (torch.rand(K, device=dev) + 0.5).to(bf)
        for M in (1, 2, 4, 8, 16):
            x = torch.randn(M, K, device=dev).to(bf); r = torch.randn(M, K, device=dev).to(bf)
            t = bench(lambda i: gemm_nt_norm(x, g, 1e-6, ws[i % copies], residual=r, silu=silu))
            print(f"N={N} K={K} silu={silu} M={M}: {t:.1f} us", flush=True)

main()
