"""Headline GEMM (4096 x 8192 x 8192 bf16), native kernel only, event-timed over 30 back-to-back calls x 5 rounds."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from triton_distributed_gh_amd.ops.gemm import gemm_nt
M, N, K = 4096, 8192, 8192
a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
b = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(20): gemm_nt(a, b, out=out, force_native=True)
res = []
for _ in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): gemm_nt(a, b, out=out, force_native=True)
    e.record(); torch.cuda.synchronize()
    res.append(s.elapsed_time(e) / 30 * 1e3)
print("us per call:", " ".join(f"{r:.1f}" for r in res), flush=True)
