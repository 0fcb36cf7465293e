"""GPU probe: correctness + speed of the gfx950 GEMM against torch.matmul (hipBLASLt)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from triton_distributed_gh_amd.ops.gemm import gemm_nt

def bench(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

torch.manual_seed(0)
for (M, N, K) in [(256, 256, 128), (512, 768, 256), (300, 520, 192), (4096, 8192, 8192), (1000, 1032, 4096)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    b = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
    c = gemm_nt(a, b, force_native=True)
    ref = (a.float() @ b.float().t())
    err = (c.float() - ref).abs().max().item()
    rel = err / ref.abs().max().item()
    print(f"M{M} N{N} K{K}: max abs err {err:.4f} rel {rel:.2e}", flush=True)
    assert rel < 2e-2, "mismatch"
# fp16
a = (torch.rand(512, 256, device="cuda") - 0.5).half(); b = (torch.rand(256, 256, device="cuda") - 0.5).half()
c = gemm_nt(a, b, force_native=True); ref = a.float() @ b.float().t()
print("fp16 err", (c.float() - ref).abs().max().item(), flush=True)

for (M, N, K) in [(4096, 8192, 8192), (8192, 8192, 8192), (4096, 4096, 4096), (4096, 1024, 8192), (4096, 8192, 1024)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    b = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_mine = bench(lambda: gemm_nt(a, b, out=out, force_native=True))
    t_torch = bench(lambda: torch.matmul(a, b.t(), out=out))
    fl = 2.0 * M * N * K
    print(f"M{M} N{N} K{K}: mine {t_mine*1e3:.1f} us ({fl/t_mine/1e9:.0f} TF)  torch {t_torch*1e3:.1f} us ({fl/t_torch/1e9:.0f} TF)", flush=True)
