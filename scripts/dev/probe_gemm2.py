"""Interleaved A/B: gfx950 GEMM vs torch.matmul on the headline shape, several data scales."""
import sys, torch
sys.path.insert(0, "/root/repo")
from triton_distributed_gh_amd.ops.gemm import gemm_nt
M, N, K = 4096, 8192, 8192
def timeit(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for scale in (1.0,):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    b = ((torch.rand(N, K, device="cuda") * 2 - 1) * scale).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f1 = lambda: gemm_nt(a, b, out=out, force_native=True, variant=1)
    f3 = lambda: gemm_nt(a, b, out=out, force_native=True, variant=2)
    f2 = lambda: torch.matmul(a, b.t(), out=out)
    for _ in range(200): f1(); f2()          # ~0.15 s warm
    res = {"mine": [], "v2": [], "torch": []}
    for rnd in range(6):
        res["mine"].append(timeit(f1, 50)); res["v2"].append(timeit(f3, 50)); res["torch"].append(timeit(f2, 50))
    fl = 2.0 * M * N * K
    for k, v in res.items():
        v = sorted(v); med = v[len(v)//2]
        print(f"scale {scale}: {k:6s} median {med:.1f} us ({fl/med/1e6:.0f} TF) min {v[0]:.1f} max {v[-1]:.1f}", flush=True)
c1 = gemm_nt(a, b, force_native=True, variant=2); ref = a.float() @ b.float().t()
print("v2 rel err", ((c1.float() - ref).abs().max() / ref.abs().max()).item())
