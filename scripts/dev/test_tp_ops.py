"""Multi-process functional test of AG-GEMM / GEMM-RS (works with several ranks on ONE GPU)."""
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import torch.distributed as dist
from triton_distributed_gh_amd.utils import initialize_distributed, finalize_distributed
from triton_distributed_gh_amd.ops.ag_gemm import create_ag_gemm_intra_node_context, ag_gemm_intra_node
from triton_distributed_gh_amd.ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node

env = initialize_distributed(heap_bytes=1 << 29)
r, w = env.rank, env.world_size
dev = env.device
ctas = max(8, 224 // w) if env.shared_gpu else 0
producer = os.environ.get("PRODUCER", "copy_engine")

def gather_cpu(t):
    outs = [torch.empty_like(t.cpu()) for _ in range(w)]
    dist.all_gather(outs, t.cpu(), group=env.cpu_group)
    return outs

def run_ag(M, N, K, iters=3, chunk=256):
    ctx = create_ag_gemm_intra_node_context(M, N // w, K, torch.bfloat16, torch.bfloat16, r, w, M_PER_CHUNK=chunk, num_ctas=ctas, producer=producer)
    for it in range(iters):
        a = (torch.rand(M // w, K, device=dev) * 2 - 1).bfloat16()
        b = (torch.rand(N // w, K, device=dev) * 2 - 1).bfloat16()
        if it == 1 and r == 1:
            time.sleep(0.3)   # straggler
        c = ag_gemm_intra_node(a, b, False, ctx)
        t0 = time.time()
        torch.cuda.synchronize()
        if time.time() - t0 > 1.0: print(f"[r{r}] AG it {it} sync took {time.time()-t0:.2f}s", flush=True)
        if int(ctx.scratch[4]) != 0:
            print(f"[r{r}] AG it {it} TIMEOUT flags0 {ctx.flags[0][r][:12].tolist()} flags1 {ctx.flags[1][r][:12].tolist()} epoch {ctx.call_count}", flush=True)
        a_full = torch.cat(gather_cpu(a)).to(dev)
        ref = a_full.float() @ b.float().t()
        err = (c.float() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-2, f"ag_gemm mismatch rank {r} it {it}: {err}"
    dist.barrier(group=env.cpu_group)
    if r == 0: print(f"AG-GEMM M{M} N{N} K{K} chunk{chunk} OK", flush=True)

def run_rs(M, N, K, iters=3, chunk=256):
    ctx = create_gemm_rs_intra_node_context(M, N, torch.bfloat16, r, w, M_PER_CHUNK=chunk, num_ctas=ctas)
    for it in range(iters):
        a = (torch.rand(M, K // w, device=dev) * 2 - 1).bfloat16()
        b = (torch.rand(N, K // w, device=dev) * 2 - 1).bfloat16()
        if it == 1 and r == 0:
            time.sleep(0.3)
        out = gemm_rs_intra_node(a, b, ctx)
        torch.cuda.synchronize()
        ctx.check_error()
        part = (a.float() @ b.float().t()).bfloat16()
        parts = gather_cpu(part)
        ref = sum(p.float() for p in parts)[r * (M // w):(r + 1) * (M // w)].to(dev)
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-2, f"gemm_rs mismatch rank {r} it {it}: {err}"
    dist.barrier(group=env.cpu_group)
    if r == 0: print(f"GEMM-RS M{M} N{N} K{K} chunk{chunk} OK", flush=True)

only = os.environ.get("ONLY", "")
run_ag(2048, 2048, 512, iters=6)
if only == "ag1": finalize_distributed(); sys.exit(0)
run_ag(1024, 1024, 256, chunk=128)
run_ag(64 * w, 512 * w, 256)          # small M: straddle mode
run_ag(4096, 8192, 1024, iters=2)
run_rs(2048, 1024, 512 * w)
run_rs(1024 * w, 2048, 256 * w, chunk=512)
run_rs(64 * w, 512, 256 * w)
run_rs(4096, 8192, 128 * w, iters=2)
finalize_distributed()
if r == 0: print("ALL OK", flush=True)
