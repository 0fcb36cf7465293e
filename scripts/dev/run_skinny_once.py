"""Decode GEMM M=1, 12800 x 8192 bf16: a few calls of the native kernel or torch.matmul (for rocprofv3 --pmc)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from triton_distributed_gh_amd.ops.gemm import gemm_nt
N, K = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (12800, 8192)
a = (torch.rand(1, K, device="cuda") - 0.5).bfloat16()
b = (torch.rand(N, K, device="cuda") - 0.5).bfloat16()
out = torch.empty(1, N, device="cuda", dtype=torch.bfloat16)
which = sys.argv[1] if len(sys.argv) > 1 else "mine"
for _ in range(12):
    if which == "mine": gemm_nt(a, b, out=out, force_native=True)
    else: torch.matmul(a, b.t(), out=out)
torch.cuda.synchronize()
