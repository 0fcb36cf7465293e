"""gemm_nt against torch.matmul on Qwen3-8B layer shapes + LM head at prefill-chunk row counts (cold weights: copies
rotated), us per call from HIP-graph replays."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.ops.gemm import gemm_nt
from triton_distributed_gh_amd.utils.perf import graph_bench
dev, bf = "cuda", torch.bfloat16
FAMILIES = {
    "qwen3-8b": [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 24576), ("down", 12288, 4096), ("lm_head", 4096, 151936)],
    "qwen3-32b": [("qkv", 5120, 10240), ("o", 8192, 5120), ("gate_up", 5120, 51200), ("down", 25600, 5120)],
    "llama3-70b": [("qkv", 8192, 10240), ("o", 8192, 8192), ("gate_up", 8192, 57344), ("down", 28672, 8192)],
    "llama3-70b-tp8": [("qkv", 8192, 1280), ("o", 1024, 8192), ("gate_up", 8192, 7168), ("down", 3584, 8192)],
}
shapes = FAMILIES[os.environ.get("FAMILY", "qwen3-8b")]
for name, K, N in shapes:
    copies = max(2, min(8, int(700e6 / (N * K * 2)) + 1))
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(copies)]
    line = []
    for M in [int(v) for v in os.environ.get("ROWS", "192,256,384,512,768,1024,1536,2048").split(",")]:
        x = torch.randn(M, K, device=dev).to(bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        res = []
        for impl in ("native", "torch"):
            it = [0]
            def f():
                i = it[0]; it[0] += 1
                (gemm_nt(x, ws[i % copies], out=out) if impl == "native" else torch.matmul(x, ws[i % copies].t(), out=out))
            f()
            res.append(graph_bench(f, n=copies * 2))
        line.append(f"M={M}: {res[0]:.1f}/{res[1]:.1f}" + ("*" if res[0] > 1.03 * res[1] else ""))
    print(f"{name} {N}x{K} native/torch: " + "  ".join(line), flush=True)
    del ws
