"""Headline GEMM shape: the native kernel and torch.matmul, a few calls each (for counter passes)."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from triton_distributed_gh_amd.ops.gemm import gemm_nt  # noqa: E402

M, N, K = 4096, 8192, 8192
a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
b = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
gm = int(sys.argv[1]) if len(sys.argv) > 1 else None
for _ in range(6):
    gemm_nt(a, b, out=out, force_native=True, group_m=gm)
    torch.matmul(a, b.t(), out=out)
torch.cuda.synchronize()
