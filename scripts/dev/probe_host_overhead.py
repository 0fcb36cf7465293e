"""Host-side cost per fused-op call (tiny shapes, so the GPU is never the bottleneck)."""
import os, sys, time, cProfile, pstats, io
os.environ["TDIST_FORCE_FUSED"] = "1"
os.environ["GPU_MAX_HW_QUEUES"] = "24"
sys.path.insert(0, "/root/repo")
import torch
from triton_distributed_gh_amd.ops.ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context
from triton_distributed_gh_amd.ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node
from triton_distributed_gh_amd.utils.testing import VirtualCluster
vc = VirtualCluster(2, heap_bytes=1 << 26)
M, N, K = 512, 512, 256
dt = torch.bfloat16
ag = vc.collective(lambda r, env: create_ag_gemm_intra_node_context(M, N // 2, K, dt, dt, r, 2, env=env))
rs = vc.collective(lambda r, env: create_gemm_rs_intra_node_context(M, N, dt, r, 2, env=env))
a = [(torch.rand(M // 2, K, device="cuda") - 0.5).to(dt) for _ in range(2)]
b = [(torch.rand(N // 2, K, device="cuda") - 0.5).to(dt) for _ in range(2)]
a2 = [(torch.rand(M, K // 2, device="cuda") - 0.5).to(dt) for _ in range(2)]
b2 = [(torch.rand(N, K // 2, device="cuda") - 0.5).to(dt) for _ in range(2)]
def host_time(fn, n=300):
    for _ in range(20): vc.run(fn, sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): vc.run(fn, sync=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n / 2 * 1e6, (t2 - t0) / n / 2 * 1e6
print("ag_gemm  host us/call %.1f  (incl. drain %.1f)" % host_time(lambda r, env: ag_gemm_intra_node(a[r], b[r], False, ag[r])))
print("gemm_rs  host us/call %.1f  (incl. drain %.1f)" % host_time(lambda r, env: gemm_rs_intra_node(a2[r], b2[r], rs[r])))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): vc.run(lambda r, env: ag_gemm_intra_node(a[r], b[r], False, ag[r]), sync=False)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3000])
