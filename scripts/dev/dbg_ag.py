import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import torch.distributed as dist
from triton_distributed_gh_amd.utils import initialize_distributed, finalize_distributed
from triton_distributed_gh_amd.ops.ag_gemm import create_ag_gemm_intra_node_context, ag_gemm_intra_node
from triton_distributed_gh_amd.ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node
env = initialize_distributed(heap_bytes=1 << 28)
r, w = env.rank, env.world_size
os.makedirs("gpurun_out", exist_ok=True)
log = open(f"gpurun_out/dbg_rank{r}.log", "w")
def P(*a):
    print(f"[r{r}]", *a, flush=True); print(*a, file=log, flush=True)
dev = env.device
ctas = 224 // w
M, N, K = [int(x) for x in os.environ.get("MNK", "1024,1024,256").split(",")]
which = os.environ.get("WHICH", "ag")
if which == "ag":
    ctx = create_ag_gemm_intra_node_context(M, N // w, K, torch.bfloat16, torch.bfloat16, r, w, num_ctas=ctas)
    P("ctx created")
    a = (torch.rand(M // w, K, device=dev) * 2 - 1).bfloat16()
    b = (torch.rand(N // w, K, device=dev) * 2 - 1).bfloat16()
    c = ag_gemm_intra_node(a, b, False, ctx)
    P("launched")
    torch.cuda.synchronize()
    P("synced; err word", int(ctx.scratch[4]), "flags", ctx.flags[0][r][:8].tolist())
else:
    ctx = create_gemm_rs_intra_node_context(M, N, torch.bfloat16, r, w, num_ctas=ctas)
    P("ctx created")
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    b = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    out = gemm_rs_intra_node(a, b, ctx)
    P("launched")
    torch.cuda.synchronize()
    P("synced; err", int(ctx.scratch[-4]), "flags", ctx.flags[0][r][:8].tolist(), "counters", ctx.scratch[:12].tolist())
dist.barrier(group=env.cpu_group)
P("done")
finalize_distributed()
