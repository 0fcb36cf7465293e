"""A few launches of gemm_nt_tiles at the headline shape for rocprofv3: run_tile_gemm_once.py <bf16|fp8|bf8|int8> [bm bn]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from triton_distributed_gh_amd.ops import gemm as G  # noqa: E402

dt = {"bf16": torch.bfloat16, "fp8": torch.float8_e4m3fn, "bf8": torch.float8_e5m2, "int8": torch.int8}[sys.argv[1]]
bm, bn = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 256)
M, N, K = 4096, 8192, 8192
if dt == torch.int8:
    a = torch.randint(-127, 128, (M, K), device="cuda", dtype=dt)
    b = torch.randint(-127, 128, (N, K), device="cuda", dtype=dt)
else:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt)
    b = (torch.rand(N, K, device="cuda") * 2 - 1).to(dt)
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(12):
    G.gemm_nt_tiles(a, b, o, bm=bm, bn=bn)
torch.cuda.synchronize()
