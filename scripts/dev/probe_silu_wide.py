"""gate_up + SiLU at 8 / 16 rows: the fused kernel's forms (TDIST_SILU_WIDE=0 | 48 | 88), cold weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.ops.gemm import gemm_nt_silu_mul
from triton_distributed_gh_amd.utils.perf import graph_bench
dev, bf = "cuda", torch.bfloat16
for name, I, K in [("qwen3-8b", 12288, 4096), ("qwen3-32b", 25600, 5120), ("llama3-70b", 28672, 8192), ("qwen3-14b", 17408, 5120)]:
    copies = max(2, min(8, int(700e6 / (2 * I * K * 2)) + 1))
    ws = [(torch.randn(2 * I, K, device=dev) * 0.02).to(bf) for _ in range(copies)]
    res = []
    for M in (8, 12, 16):
        x = torch.randn(M, K, device=dev).to(bf)
        it = [0]
        def f():
            i = it[0]; it[0] += 1
            return gemm_nt_silu_mul(x, ws[i % copies])
        y = f()
        g = x.float() @ ws[(it[0] - 1) % copies].float().t()
        ref = torch.nn.functional.silu(g[:, :I].to(bf).float()) * g[:, I:].to(bf).float()
        err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
        res.append(f"M={M}: {graph_bench(f, n=copies * 2):.1f}" + ("" if err < 2e-2 else f" (ERR {err:.1e})"))
    print(f"{name} I={I} K={K} [wide={os.environ.get('TDIST_SILU_WIDE', '0')}]: " + "  ".join(res), flush=True)
    del ws
