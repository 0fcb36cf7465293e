import sys, torch
sys.path.insert(0, "/root/repo")
from triton_distributed_gh_amd.ops.gemm import gemm_nt
def timeit(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N, K) in [(4096, 8192, 8192), (4096, 4096, 8192), (4096, 8192, 4096)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    b = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fns = {"v2": lambda: gemm_nt(a, b, out=out, force_native=True, variant=2),
           "v2-nostore": lambda: gemm_nt(a, b, out=out, force_native=True, variant=2 + 256),
           "v2-g4": lambda: gemm_nt(a, b, out=out, force_native=True, variant=2, group_m=4),
           "v2-g16": lambda: gemm_nt(a, b, out=out, force_native=True, variant=2, group_m=16),
           "torch": lambda: torch.matmul(a, b.t(), out=out)}
    for _ in range(100):
        for f in fns.values(): f()
    res = {k: [] for k in fns}
    for rnd in range(5):
        for k, f in fns.items(): res[k].append(timeit(f, 40))
    fl = 2.0 * M * N * K
    print(f"M{M} N{N} K{K}: " + "  ".join(f"{k} {sorted(v)[2]:.1f}us/{fl/sorted(v)[2]/1e6:.0f}TF" for k, v in res.items()), flush=True)
