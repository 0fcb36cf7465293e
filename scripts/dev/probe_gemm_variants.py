"""Interleaved A/B of GEMM schedule variants vs torch.matmul on the headline shape (random data).
usage: probe_gemm_variants.py [variant ...]   (variant = integer passed to gemm_nt; default set below)"""
import sys

import torch

sys.path.insert(0, "/root/repo")
from triton_distributed_gh_amd.ops.gemm import gemm_nt  # noqa: E402

M, N, K = 4096, 8192, 8192
variants = [int(v, 0) for v in sys.argv[1:]] or [2, 5, 6, 7, 8, 3 | (3 << 16)]


def timeit(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
b = ((torch.rand(N, K, device="cuda") * 2 - 1)).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ref = a.float() @ b.float().t()
fns = {"torch": lambda: torch.matmul(a, b.t(), out=out)}
for v in variants:
    c = gemm_nt(a, b, force_native=True, variant=v)
    torch.cuda.synchronize()
    err = ((c.float() - ref).abs().max() / ref.abs().max()).item()
    bad = int(((c.float() - ref).abs() > 1e-2 * ref.abs() + 0.5).sum())
    print(f"variant {v:#x}: rel err {err:.2e}, elements off {bad}", flush=True)
    assert err < 1e-2 and bad == 0, f"variant {v} is wrong"
    fns[f"v{v:#x}"] = (lambda vv: (lambda: gemm_nt(a, b, out=out, force_native=True, variant=vv)))(v)
del ref
for _ in range(30):
    for f in fns.values():
        f()
torch.cuda.synchronize()
res = {k: [] for k in fns}
for rnd in range(8):
    for k, f in fns.items():
        res[k].append(timeit(f, 40))
fl = 2.0 * M * N * K
base = sorted(res["torch"])[len(res["torch"]) // 2]
for k, v in res.items():
    v = sorted(v)
    med = v[len(v) // 2]
    print(f"{k:10s} median {med:7.1f} us ({fl / med / 1e6:5.0f} TF) min {v[0]:7.1f} max {v[-1]:7.1f}  vs torch {base / med:.3f}x", flush=True)
