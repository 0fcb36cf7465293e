"""Plain decode GEMMs at 2 .. 16 rows (Qwen3-8B layer shapes + LM head), cold weights, through gemm_nt; the kernel is
picked per process by TDIST_SKINNY128 (0 = 64-byte-run MFMA kernel; WN NW digits = 128-byte-run kernel) and
TDIST_DECODE_SPLITK_MIN_M (17 = never the split-K mid-M kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.ops.gemm import gemm_nt
from triton_distributed_gh_amd.utils.perf import graph_bench
dev, bf = "cuda", torch.bfloat16
tag = f"s128={os.environ.get('TDIST_SKINNY128', '0')} splitk_min_m={os.environ.get('TDIST_DECODE_SPLITK_MIN_M', '2')}"
for name, N, K in [("qkv", 6144, 4096), ("o_proj", 4096, 4096), ("down", 4096, 12288), ("gate_up", 24576, 4096), ("lm_head", 151936, 4096)]:
    copies = max(2, min(24, int(700e6 / (N * K * 2)) + 1))
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(copies)]
    res = []
    for M in (2, 4, 8, 16):
        x = torch.randn(M, K, device=dev).to(bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        it = [0]
        def f():
            i = it[0]; it[0] += 1
            gemm_nt(x, ws[i % copies], out=out)
        f()
        ref = x.float() @ ws[0].float().t()
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        res.append(f"M={M}: {graph_bench(f, n=copies):.1f}" + ("" if err < 2e-2 else f" (ERR {err:.1e})"))
    print(f"{name} [{tag}]: " + "  ".join(res), flush=True)
    del ws
