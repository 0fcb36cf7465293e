"""time(K) at fixed M=4096, N=8192: slope = main-loop cost per K, intercept = per-launch / per-tile overhead."""
import sys

import torch

sys.path.insert(0, "/root/repo")
from triton_distributed_gh_amd.ops.gemm import gemm_nt  # noqa: E402

M, N = 4096, 8192
variants = [int(v, 0) for v in sys.argv[1:]] or [2, 3 | (3 << 16)]


def timeit(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


rows = {}
for K in (1024, 2048, 4096, 8192, 16384):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    b = ((torch.rand(N, K, device="cuda") * 2 - 1)).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fns = {"torch": lambda: torch.matmul(a, b.t(), out=out)}
    for v in variants:
        fns[f"v{v:#x}"] = (lambda vv: (lambda: gemm_nt(a, b, out=out, force_native=True, variant=vv)))(v)
    for _ in range(20):
        for f in fns.values():
            f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    iters = max(10, int(40 * 8192 / K))
    for rnd in range(6):
        for k, f in fns.items():
            res[k].append(timeit(f, iters))
    line = f"K={K:6d} "
    for k, v in res.items():
        med = sorted(v)[len(v) // 2]
        rows.setdefault(k, []).append((K, med))
        line += f" {k} {med:7.1f} us"
    print(line, flush=True)
for k, pts in rows.items():
    (k0, t0), (k1, t1) = pts[1], pts[-1]
    slope = (t1 - t0) / (k1 - k0)
    print(f"{k:10s} slope {slope * 1024:6.2f} us per 1024 of K, intercept {t0 - slope * k0:6.2f} us "
          f"(from K={k0} and K={k1})", flush=True)
