"""Register-staged four-wave GEMM (variant 12) vs the eight-wave persistent kernel (variant 3) and torch.matmul:
correctness on full and ragged shapes (elementwise), then interleaved timing at the headline shape."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from triton_distributed_gh_amd.ops.gemm import gemm_nt  # noqa: E402


def timeit(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


ok = True
for (M, N, K) in [(256, 256, 128), (512, 768, 256), (300, 520, 192), (4096, 8192, 1024), (1000, 8200, 512), (4096, 1024, 8192)]:
    for dt in (torch.bfloat16, torch.float16):
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt)
        b = (torch.rand(N, K, device="cuda") * 2 - 1).to(dt)
        out = torch.full((M, N), 7.0, device="cuda", dtype=dt)
        ref = a.float() @ b.float().t()
        for variant in (12, 13):
            out.fill_(7.0)
            gemm_nt(a, b, out=out, force_native=True, variant=variant)
            torch.cuda.synchronize()
            good = torch.allclose(out.float(), ref, atol=0.15 * (K / 1024) ** 0.5, rtol=2e-2)
            err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
            ok &= good
            print(f"v{variant} {M}x{N}x{K} {dt}: rel err {err:.2e} {'ok' if good else 'FAIL'}", flush=True)
if not ok:
    sys.exit(1)

M, N, K = 4096, 8192, 8192
a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
b = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
fns = {"v3": lambda: gemm_nt(a, b, out=out, force_native=True, variant=3 | (3 << 16)),
       "v12": lambda: gemm_nt(a, b, out=out, force_native=True, variant=12),
       "v13": lambda: gemm_nt(a, b, out=out, force_native=True, variant=13),
       "torch": lambda: torch.matmul(a, b.t(), out=out)}
for _ in range(40):
    for f in fns.values():
        f()
res = {k: [] for k in fns}
for rnd in range(6):
    for k, f in fns.items():
        res[k].append(timeit(f, 50))
fl = 2.0 * M * N * K
for k, v in res.items():
    v = sorted(v)
    med = v[len(v) // 2]
    print(f"{k:6s} median {med:.1f} us ({fl / med / 1e6:.0f} TF) min {v[0]:.1f} max {v[-1]:.1f}", flush=True)
