"""Decode GEMMs (1 - 16 rows) with cold weights: gemm_nt (this package's choice) against torch.matmul (hipBLASLt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.ops.gemm import gemm_nt
from triton_distributed_gh_amd.utils.perf import graph_bench
dev, bf = "cuda", torch.bfloat16
for name, N, K in [("qkv", 6144, 4096), ("o_proj", 4096, 4096), ("down", 4096, 12288), ("gate_up", 24576, 4096), ("lm_head", 151936, 4096)]:
    copies = max(2, min(24, int(700e6 / (N * K * 2)) + 1))
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(copies)]
    res = []
    for M in (1, 2, 4, 8, 16):
        x = torch.randn(M, K, device=dev).to(bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        it = [0]
        def ours():
            i = it[0]; it[0] += 1
            gemm_nt(x, ws[i % copies], out=out)
        def libf():
            i = it[0]; it[0] += 1
            torch.matmul(x, ws[i % copies].t(), out=out)
        res.append(f"M={M}: {graph_bench(ours, n=copies):.1f} / {graph_bench(libf, n=copies):.1f}")
    print(name, "(ours / lib)", "  ".join(res), flush=True)
    del ws
