"""Alternating-operand GEMM timing (cold Infinity Cache): variants / group_m vs torch.matmul."""
import sys, torch
sys.path.insert(0, "/root/repo")
from triton_distributed_gh_amd.ops.gemm import gemm_nt
M, N, K = 4096, 8192, 8192
dt = torch.bfloat16
sets = [((torch.rand(M, K, device="cuda") * 2 - 1).to(dt), ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(dt)) for _ in range(3)]
outs = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(3)]

def bench(fn, iters=60):
    for i in range(9): fn(i % 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fn(i % 3)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def same(fn, iters=60):
    for i in range(9): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fn(0)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for rep in range(2):
    print("torch      alt %.1f same %.1f" % (bench(lambda i: torch.matmul(sets[i][0], sets[i][1].t(), out=outs[i])), same(lambda i: torch.matmul(sets[i][0], sets[i][1].t(), out=outs[i]))), flush=True)
    for var in (1, 2):
        for gm in (4, 8, 16):
            f = lambda i: gemm_nt(sets[i][0], sets[i][1], out=outs[i], group_m=gm, variant=var, force_native=True)
            print("var %d gm %2d alt %.1f same %.1f" % (var, gm, bench(f), same(f)), flush=True)
