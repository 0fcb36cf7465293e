"""GEMM + AllReduce: fused low-latency form against gemm_nt + all_reduce of this package, w virtual ranks on
one GPU (every rank's kernels run concurrently; host-timed, gate-released).

    python scripts/dev/probe_gemm_ar.py 8 128 5120 25600
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from triton_distributed_gh_amd.ops.allreduce import AllReduceMethod, all_reduce, create_allreduce_ctx  # noqa: E402
from triton_distributed_gh_amd.ops.gemm import gemm_nt  # noqa: E402
from triton_distributed_gh_amd.ops.gemm_ar import create_gemm_ar_context, gemm_allreduce_op  # noqa: E402
from triton_distributed_gh_amd.utils.testing import VirtualCluster  # noqa: E402


def main():
    w, M, N, K = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 128, 5120, 25600)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(32, 2 * w + 4)))
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    dt = torch.bfloat16
    ctxs = vc.collective(lambda r, env: create_gemm_ar_context(None, r, w, w, M, N, dt, num_ctas=max(8, 200 // w), env=env))
    arc = vc.collective(lambda r, env: create_allreduce_ctx(M * N * 2 + (1 << 20), r, w, env=env))
    a = [(torch.rand(M, K // w, device="cuda") - 0.5).to(dt) for _ in range(w)]
    b = [(torch.rand(N, K // w, device="cuda") - 0.5).to(dt) for _ in range(w)]
    tmp = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(w)]

    def fused(r, env):
        return gemm_allreduce_op(ctxs[r], a[r], b[r], copy_to_local=False)

    def unfused(r, env):
        gemm_nt(a[r], b[r], out=tmp[r])
        return all_reduce(tmp[r], AllReduceMethod.TwoShot if M * N * 2 > (256 << 10) else AllReduceMethod.OneShot, arc[r])

    def gemm_only(r, env):
        return gemm_nt(a[r], b[r], out=tmp[r])

    def med(fn, n=25):
        for _ in range(5):
            vc.run(fn)
        base = sorted(vc.timed_run(lambda r, env: None) for _ in range(15))[7]
        t = sorted(vc.timed_run(fn) for _ in range(n))
        return (t[len(t) // 2] - base) * 1e3

    f, u, g = med(fused), med(unfused), med(gemm_only)
    for c in ctxs:
        c.check_error()
    print(f"w={w} M={M} N={N} K={K} (K/w={K // w}): fused gemm+ar {f:7.1f} us | gemm_nt + all_reduce {u:7.1f} us | "
          f"gemm_nt alone {g:7.1f} us  (all {w} ranks concurrently on one GPU)", flush=True)
    vc.close()


if __name__ == "__main__":
    main()
