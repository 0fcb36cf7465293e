"""Plain GEMM 4096x8192x8192 bf16: group_m x schedule variant sweep, interleaved with torch.matmul."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from triton_distributed_gh_amd.ops.gemm import gemm_nt

M, N, K = 4096, 8192, 8192
a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
b = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

def bench(fn, iters=40, warmup=10):
    for _ in range(warmup): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for _ in range(3):
    bench(lambda: torch.matmul(a, b.t(), out=out))
for rep in range(2):
    print(f"torch: {bench(lambda: torch.matmul(a, b.t(), out=out)):.1f} us", flush=True)
    for variant in (2, 1):
        for gm in (1, 2, 4, 8, 16):
            t = bench(lambda: gemm_nt(a, b, out=out, group_m=gm, variant=variant, force_native=True))
            print(f"variant {variant} group_m {gm:2d}: {t:.1f} us", flush=True)
