"""Plain GEMM shapes with few 256x256 tiles: split-K kernel vs one item per tile vs torch.matmul (hipBLASLt)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from triton_distributed_gh_amd.ops import gemm as G  # noqa: E402


def bench(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(f"{'M':>6} {'N':>6} {'K':>6} {'tiles':>5} | {'split-K us':>10} {'1 item/tile us':>14} {'torch us':>9}")
for M, N, K in [(2048, 2048, 8192), (2048, 4096, 4096), (2048, 1024, 8192), (1536, 4096, 4096), (4096, 1024, 8192),
                (2048, 6144, 4096), (3072, 2048, 2048)]:
    a = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
    b = (torch.rand(N, K, device="cuda") - 0.5).bfloat16()
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tiles = -(-M // 256) * -(-N // 256)
    t_s = bench(lambda: G.gemm_nt_splitk(a, b, o))
    t_1 = bench(lambda: G.gemm_nt(a, b, o, force_native=True))
    t_t = bench(lambda: torch.matmul(a, b.t(), out=o))
    try:
        t_m = bench(lambda: G.gemm_nt_mid(a, b, o))
        ref = a.float() @ b.float().t()
        err = (o.float() - ref).abs().max().item() / ref.abs().max().item()
    except Exception as exc:  # noqa: BLE001
        t_m, err = float("nan"), str(exc)[:60]
    print(f"{M:>6} {N:>6} {K:>6} {tiles:>5} | {t_s:>10.1f} {t_1:>14.1f} {t_t:>9.1f} | mid {t_m:>7.1f} (rel err {err})", flush=True)
