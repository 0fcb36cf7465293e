"""Mid-M GEMM (M = 32 / 64 / 128) on Qwen3-8B layer shapes with COLD weights (rotating copies inside a HIP graph)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.ops.gemm import gemm_nt_mid
from triton_distributed_gh_amd.utils.perf import graph_bench
dev, bf = "cuda", torch.bfloat16
for name, N, K in [("qkv", 6144, 4096), ("o_proj", 4096, 4096), ("down", 4096, 12288), ("gate_up", 24576, 4096)]:
    copies = max(2, min(24, int(700e6 / (N * K * 2)) + 1))
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(copies)]
    res = []
    for M in (32, 64, 128):
        x = torch.randn(M, K, device=dev).to(bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        it = [0]
        def f():
            i = it[0]; it[0] += 1
            gemm_nt_mid(x, ws[i % copies], out)
        res.append(f"M={M}: {graph_bench(f, n=copies):.1f}")
    print(name, "  ".join(res), flush=True)
    del ws
