"""Compute-side cost of GEMM + AllReduce for ONE rank of w: rank 0 of a w-rank virtual cluster runs alone on the GPU
with its peers' flags pre-set (nothing waits; partials and broadcasts land in local HBM), fused against
gemm_nt + all_reduce of this package under the same conditions.  HIP events, medians.

    python scripts/dev/probe_gemm_ar_solo.py [w] [M N K_total]      default: 8  2048 5120 25600 (the reference's "MLP M=2048" row)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from triton_distributed_gh_amd.ops import allreduce as AR  # noqa: E402
from triton_distributed_gh_amd.ops.gemm import gemm_nt  # noqa: E402
from triton_distributed_gh_amd.ops.gemm_ar import create_gemm_ar_context, gemm_allreduce_op  # noqa: E402
from triton_distributed_gh_amd.utils.testing import VirtualCluster  # noqa: E402


def med(fn, pre, n=24):
    ts = []
    for _ in range(n):
        pre()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[4:])
    return ts[len(ts) // 2]


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    shapes = [tuple(int(x) for x in sys.argv[2:5])] if len(sys.argv) >= 5 else [(2048, 5120, 25600), (1024, 5120, 25600),
                                                                                (4096, 5120, 25600), (2048, 4096, 32768)]
    dt = torch.bfloat16
    for M, N, K in shapes:
        vc = VirtualCluster(w, heap_bytes=1 << 30)
        ctxs = vc.collective(lambda r, env: create_gemm_ar_context(None, r, w, w, M, N, dt, num_ctas=256, env=env))
        arc = vc.collective(lambda r, env: AR.create_allreduce_ctx(M * N * 4 + (4 << 20), r, w, env=env))
        c, ac = ctxs[0], arc[0]
        a = (torch.rand(M, K // w, device="cuda") - 0.5).to(dt)
        b = (torch.rand(N, K // w, device="cuda") - 0.5).to(dt)
        tmp = torch.empty(M, N, device="cuda", dtype=dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)

        def pre_fused():
            n = c.call_count                     # calls so far = the device's call number
            c.flags[n & 1][0].fill_(n % 4093 + 1)

        def pre_ar():
            ac.flags[ac.call_count & 1][0].fill_(ac.call_count % AR.EPOCH_MOD + 1)

        with torch.cuda.stream(vc.streams[0]):
            t_f = med(lambda: gemm_allreduce_op(c, a, b, copy_to_local=False), pre_fused)
            t_g = med(lambda: gemm_nt(a, b, out=tmp), lambda: None)
            t_a = med(lambda: AR.all_reduce(tmp, AR.AllReduceMethod.TwoShot, ac, output=out), pre_ar)

            def both():
                gemm_nt(a, b, out=tmp)
                AR.all_reduce(tmp, AR.AllReduceMethod.TwoShot, ac, output=out)

            t_u = med(both, pre_ar)
        c.check_error()
        ac.check_error()
        print(f"w={w} M={M} N={N} K/w={K // w}: fused gemm+ar {t_f:7.1f} us | gemm_nt + all_reduce {t_u:7.1f} us "
              f"(gemm_nt {t_g:6.1f} + all_reduce {t_a:6.1f}) | fused is {100 * (1 - t_f / t_u):5.1f} % under", flush=True)
        vc.close()


if __name__ == "__main__":
    main()
