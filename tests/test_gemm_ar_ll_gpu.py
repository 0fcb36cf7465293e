"""Low-latency GEMM + AllReduce (mid-M tile body, two-shot per block, device-side call numbering):
element-wise numerics on virtual ranks, HIP-graph replay."""
import pytest
import torch

from triton_distributed_gh_amd.ops.gemm_ar import create_gemm_ar_context, gemm_allreduce_op
from triton_distributed_gh_amd.utils.testing import VirtualCluster

pytestmark = pytest.mark.gpu


def _reference(a, b, w):
    parts = [(a[r].float() @ b[r].float().t()).bfloat16().float() for r in range(w)]
    return sum(parts), sum(p.abs() for p in parts)


def _assert_close(out, ref, mag, what):
    bad = (out.float() - ref).abs() > 1e-3 + 2e-2 * mag
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} elements off"


@pytest.mark.parametrize("w,M,N,K", [(2, 1, 1024, 512), (2, 64, 2048, 1024), (4, 16, 1536, 2048), (4, 128, 5120, 1024),
                                     (8, 128, 5120, 3200 * 8), (8, 200, 1280, 2048), (8, 33, 136, 1024)])
def test_ll_gemm_allreduce_elementwise(w, M, N, K):
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        ctxs = vc.collective(lambda r, env: create_gemm_ar_context(None, r, w, w, max(M, 64), N, torch.bfloat16,
                                                                   num_ctas=max(8, 200 // w), env=env))
        assert ctxs[0].ll is not None
        for it in range(4):
            a = [(torch.rand(M, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            b = [(torch.rand(N, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            torch.cuda.synchronize()
            outs = vc.run(lambda r, env: gemm_allreduce_op(ctxs[r], a[r], b[r]))
            ref, mag = _reference(a, b, w)
            for r in range(w):
                ctxs[r].check_error()
                _assert_close(outs[r], ref, mag, f"iteration {it} rank {r}")
                assert torch.equal(outs[r], outs[0])
    finally:
        vc.close()


@pytest.mark.parametrize("w,M", [(2, 16), (4, 64)])
def test_ll_gemm_allreduce_graph_replay(w, M):
    """The launch is the same every call (call number lives on the device): capture once, replay with fresh
    inputs, mix in an eager call, and check every output element each time."""
    N, K = 2048, 2048
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        ctxs = vc.collective(lambda r, env: create_gemm_ar_context(None, r, w, w, M, N, torch.bfloat16,
                                                                   num_ctas=max(8, 200 // w), env=env))
        a = [torch.zeros(M, K // w, device="cuda", dtype=torch.bfloat16) for _ in range(w)]
        b = [(torch.rand(N, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
        outs = [torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(w)]

        def step(r, env):
            outs[r].copy_(gemm_allreduce_op(ctxs[r], a[r], b[r]))

        vc.run(step)                                     # warm-up: allocations, module load
        graphs = []
        for r in range(w):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=vc.streams[r]):
                step(r, vc.envs[r])
            graphs.append(g)
        for it in range(7):
            for r in range(w):
                a[r].copy_((torch.rand(M, K // w, device="cuda") - 0.5).bfloat16())
                outs[r].zero_()
            torch.cuda.synchronize()
            if it == 3:
                vc.run(step)
            else:
                vc.run(lambda r, env: graphs[r].replay())
            ref, mag = _reference(a, b, w)
            for r in range(w):
                ctxs[r].check_error()
                _assert_close(outs[r], ref, mag, f"replay {it} rank {r}")
    finally:
        vc.close()
