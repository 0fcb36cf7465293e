"""Decode GEMMs of a Qwen3-8B layer + LM head, event-timed inside a HIP graph over rotating weight copies."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.ops.gemm import gemm_nt_norm, gemm_nt

def bench(fn, n, reps=5):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn(0)
        with torch.cuda.graph(g, stream=st):
            for i in range(n): fn(i)
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3

dev, bf = "cuda", torch.bfloat16
shapes = [("qkv+norm", 6144, 4096, "norm"), ("gate_up+norm+silu", 12288, 4096, "silu"), ("o_proj", 4096, 4096, ""),
          ("down", 4096, 12288, ""), ("lm_head", 151936, 4096, "")]
for name, N, K, mode in shapes:
    rows = 2 * N if mode == "silu" else N
    copies = max(2, min(24, int(700e6 / (rows * K * 2)) + 1))
    ws = [(torch.randn(rows, K, device=dev) * 0.02).to(bf) for _ in range(copies)]
    g = (torch.rand(K, device=dev) + 0.5).to(bf)
    out = []
    for M in (1, 4, 16):
        x = torch.randn(M, K, device=dev).to(bf); r = torch.randn(M, K, device=dev).to(bf)
        if mode:
            t = bench(lambda i: gemm_nt_norm(x, g, 1e-6, ws[i % copies], residual=r, silu=(mode == "silu")), copies)
        else:
            os.environ["X"] = "1"
            t = bench(lambda i: gemm_nt(x, ws[i % copies]), copies)
        out.append(f"M={M}: {t:.1f}")
    print(f"{name} {N}x{K} ({rows * K * 2 / 1e6:.0f} MB): " + "  ".join(out), flush=True)
    del ws
