"""GEMM + AllReduce, 256x256-tile form (more than ops.gemm_ar.LL_MAX_M rows): every element against the fp32 sum of
the bf16-rounded per-rank partials, results bitwise equal on all ranks, HIP-graph replay at op level, and the
parity-independent result buffer (an eager ``copy_to_local=False`` call between captured ones must not return the
previous call's numbers).

Scenarios: reference python/triton_dist/test/nvidia/test_gemm_ar.py (check loop, 8 ranks) and the graph capture of
layers/nvidia/tp_mlp.py under models/engine.py:_init_cuda_graph.
"""
import pytest
import torch

from triton_distributed_gh_amd.ops.gemm_ar import LL_MAX_M, create_gemm_ar_context, gemm_allreduce_op
from triton_distributed_gh_amd.utils.testing import VirtualCluster, assert_close_elementwise

pytestmark = pytest.mark.gpu


def _ref(a, b):
    parts = [(x.float() @ y.float().t()).bfloat16().float() for x, y in zip(a, b)]
    return sum(parts), sum(p.abs() for p in parts)


@pytest.mark.parametrize("w,M,N,K", [(2, 512, 1024, 512), (4, 1024, 2048, 1024), (8, 2048, 1280, 2048),
                                     (4, 700, 1096, 512),        # ragged rows and columns
                                     (8, 300, 5120, 1024)])      # more column tiles than row tiles
def test_tiled_gemm_allreduce(w, M, N, K):
    assert M > LL_MAX_M
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        ctxs = vc.collective(lambda r, env: create_gemm_ar_context(None, r, w, w, M, N, torch.bfloat16,
                                                                   num_ctas=max(8, 200 // w), env=env))
        for it in range(3):
            a = [(torch.rand(M, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            b = [(torch.rand(N, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            torch.cuda.synchronize()

            def step(r, env):
                if it == 1 and r == w - 1:
                    torch.cuda._sleep(20_000_000)      # straggler
                return gemm_allreduce_op(ctxs[r], a[r], b[r])

            outs = vc.run(step)
            ref, mag = _ref(a, b)
            for r in range(w):
                ctxs[r].check_error()
                assert_close_elementwise(outs[r], ref, what=f"gemm_allreduce rank {r} iteration {it}", magnitude=mag)
                assert torch.equal(outs[r], outs[0])
    finally:
        vc.close()


@pytest.mark.parametrize("w,M", [(2, 512), (4, 768)])
def test_tiled_gemm_allreduce_graph_replay(w, M):
    """One kernel node, the same every call: capture once, replay with fresh inputs; an eager call with
    copy_to_local=False in between returns THIS call's result and keeps the sequence."""
    N, K = 1024, 1024
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        ctxs = vc.collective(lambda r, env: create_gemm_ar_context(None, r, w, w, M, N, torch.bfloat16,
                                                                   num_ctas=max(8, 200 // w), env=env))
        a = [torch.zeros(M, K // w, device="cuda", dtype=torch.bfloat16) for _ in range(w)]
        b = [(torch.rand(N, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
        outs = [torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(w)]

        def step(r, env):
            outs[r].copy_(gemm_allreduce_op(ctxs[r], a[r], b[r]))

        def eager_view(r, env):
            outs[r].copy_(gemm_allreduce_op(ctxs[r], a[r], b[r], copy_to_local=False))

        vc.run(step)
        graphs = []
        for r in range(w):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=vc.streams[r]):
                step(r, vc.envs[r])
            graphs.append(g)
        for it in range(8):
            for r in range(w):
                a[r].copy_((torch.rand(M, K // w, device="cuda") - 0.5).bfloat16())
                outs[r].zero_()
            torch.cuda.synchronize()
            if it in (2, 5):                  # an odd number of replays lies between the two eager calls
                vc.run(eager_view)
            else:
                vc.run(lambda r, env: graphs[r].replay())
            torch.cuda.synchronize()
            ref, mag = _ref(a, b)
            for r in range(w):
                ctxs[r].check_error()
                assert_close_elementwise(outs[r], ref, what=f"rank {r} iteration {it}", magnitude=mag)
    finally:
        vc.close()


@pytest.mark.parametrize("w,M", [(2, 16), (4, 64)])
def test_ll_eager_view_between_replays(w, M):
    """Low-latency form, same hazard: the result of an eager copy_to_local=False call after an odd number of
    captured calls used to come from the other parity's buffer."""
    N, K = 2048, 1024
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        ctxs = vc.collective(lambda r, env: create_gemm_ar_context(None, r, w, w, M, N, torch.bfloat16,
                                                                   num_ctas=max(8, 200 // w), env=env))
        a = [torch.zeros(M, K // w, device="cuda", dtype=torch.bfloat16) for _ in range(w)]
        b = [(torch.rand(N, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
        outs = [torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(w)]

        def step(r, env):
            outs[r].copy_(gemm_allreduce_op(ctxs[r], a[r], b[r], copy_to_local=False))

        vc.run(step)
        graphs = []
        for r in range(w):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=vc.streams[r]):
                step(r, vc.envs[r])
            graphs.append(g)
        for it in range(6):
            for r in range(w):
                a[r].copy_((torch.rand(M, K // w, device="cuda") - 0.5).bfloat16())
                outs[r].zero_()
            torch.cuda.synchronize()
            if it in (1, 4):
                vc.run(step)
            else:
                vc.run(lambda r, env: graphs[r].replay())
            torch.cuda.synchronize()
            ref, mag = _ref(a, b)
            for r in range(w):
                ctxs[r].check_error()
                assert_close_elementwise(outs[r], ref, what=f"rank {r} iteration {it}", magnitude=mag)
    finally:
        vc.close()
