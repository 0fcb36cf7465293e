"""Compute-side cost of the fused TP kernels at the headline shape, one rank at a time on one GPU.

Rank 0 of a w-rank virtual cluster runs alone with the whole GPU; the arrival flags it would wait
on are pre-set to the call's epoch, so nothing waits on a peer and every push / scatter lands in
local HBM.  The result is what a rank of a real TP=w run would take if xGMI were free - the number
to compare with the plain GEMM of the same per-rank shape (ours and hipBLASLt's)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from triton_distributed_gh_amd.ops.ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context  # noqa: E402
from triton_distributed_gh_amd.ops.gemm import gemm_nt  # noqa: E402
from triton_distributed_gh_amd.ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node  # noqa: E402
from triton_distributed_gh_amd.utils.testing import VirtualCluster  # noqa: E402

M, N, K = 4096, 8192, 8192
EPOCH_MOD = 4093


def timeit(fn, n=30, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    ws = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]
    dt = torch.bfloat16
    os.environ["TDIST_FORCE_FUSED"] = "1"
    for w in ws:
        vc = VirtualCluster(w, heap_bytes=1 << 30)
        ag = vc.collective(lambda r, env: create_ag_gemm_intra_node_context(
            M, N // w, K, dt, dt, r, w, num_ctas=256, env=env, producer=os.environ.get("PROBE_PRODUCER", "kernel")))
        rs = vc.collective(lambda r, env: create_gemm_rs_intra_node_context(M, N, dt, r, w, num_ctas=256, env=env))
        a1 = (torch.rand(M // w, K, device="cuda") * 2 - 1).to(dt)
        a1f = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt)
        w1 = (torch.rand(N // w, K, device="cuda") * 2 - 1).to(dt)
        a2 = (torch.rand(M, K // w, device="cuda") * 2 - 1).to(dt)
        w2 = (torch.rand(N, K // w, device="cuda") * 2 - 1).to(dt)

        def run_ag():
            c = ag[0]
            ep = c.call_count % EPOCH_MOD + 1
            c.flags[c.call_count & 1][0].fill_(ep)
            return ag_gemm_intra_node(a1, w1, False, c)

        def run_rs():
            c = rs[0]
            ep = c.call_count % EPOCH_MOD + 1
            c.flags[c.call_count & 1][0].fill_(ep)
            return gemm_rs_intra_node(a2, w2, c)

        def fill_only():
            ag[0].flags[0][0].fill_(1)

        with torch.cuda.stream(vc.streams[0]):
            t_fill = timeit(fill_only)
            t_ag = timeit(run_ag) - t_fill
            t_rs = timeit(run_rs) - t_fill
            o1 = torch.empty(M, N // w, device="cuda", dtype=dt)
            o2 = torch.empty(M, N, device="cuda", dtype=dt)
            t_g1 = timeit(lambda: gemm_nt(a1f, w1, out=o1))
            t_g2 = timeit(lambda: gemm_nt(a2, w2, out=o2))
            t_b1 = timeit(lambda: torch.matmul(a1f, w1.t(), out=o1))
            t_b2 = timeit(lambda: torch.matmul(a2, w2.t(), out=o2))
        ag[0].check_error()
        rs[0].check_error()
        print(f"w={w}: ag_gemm {t_ag:7.1f} us (gemm_nt {t_g1:6.1f}, hipblaslt {t_b1:6.1f}) | "
              f"gemm_rs {t_rs:7.1f} us (gemm_nt {t_g2:6.1f}, hipblaslt {t_b2:6.1f})", flush=True)
        vc.close()


if __name__ == "__main__":
    main()
