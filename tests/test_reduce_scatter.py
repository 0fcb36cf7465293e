"""Stand-alone reduce-scatter vs a PyTorch fp32 sum (scenario of the reference's
test/nvidia/test_reduce_scatter.py: random partials per rank, compare with reduce_scatter_tensor)."""
import pytest
import torch

from triton_distributed_gh_amd.ops.reduce_scatter import create_reduce_scater_2d_ctx, reduce_scatter_2d_op, ring_reduce
from triton_distributed_gh_amd.utils.testing import VirtualCluster


def _run(w, M, N, dtype, backend, iters=3):
    dev = "cuda" if backend == "hip" else "cpu"
    vc = VirtualCluster(w, heap_bytes=1 << 26, backend=backend)
    try:
        ctxs = vc.collective(lambda r, env: create_reduce_scater_2d_ctx(M, N, r, w, w, dtype, env=env))
        for it in range(iters):
            xs = [(torch.rand(M, N, device=dev) * 2 - 1).to(dtype) for _ in range(w)]
            if backend == "hip":
                torch.cuda.synchronize()

            def step(r, env):
                if backend == "hip" and it == 1 and r == 0:
                    torch.cuda._sleep(10_000_000)
                return reduce_scatter_2d_op(xs[r], ctxs[r])

            outs = vc.run(step)
            ref = sum(x.float() for x in xs)
            Mr = M // w
            for r in range(w):
                ctxs[r].check_error()
                assert torch.allclose(outs[r].float(), ref[r * Mr:(r + 1) * Mr], atol=3e-2, rtol=2e-2), (r, it)
    finally:
        vc.close()


@pytest.mark.parametrize("w", [1, 2, 4])
def test_reduce_scatter_host(w):
    _run(w, 16 * w, 24, torch.float32, "host")


@pytest.mark.gpu
@pytest.mark.parametrize("w,M,N,dtype", [(2, 512, 1024, torch.bfloat16), (4, 1024, 2048, torch.bfloat16),
                                         (8, 4096, 1024, torch.float16), (8, 64, 8, torch.bfloat16)])
def test_reduce_scatter_gpu(w, M, N, dtype):
    _run(w, M, N, dtype, "hip")


@pytest.mark.gpu
@pytest.mark.parametrize("w", [2, 4])
def test_reduce_scatter_graph_replay(w):
    """The call number lives on the device and the input copy is one buffer: the staging copy and the kernel are
    captured once per rank and replayed with fresh partial sums, an eager call in between, one rank held back."""
    M, N, dtype = 512, 1024, torch.bfloat16
    vc = VirtualCluster(w, heap_bytes=1 << 26)
    try:
        ctxs = vc.collective(lambda r, env: create_reduce_scater_2d_ctx(M, N, r, w, w, dtype, env=env))
        xs = [torch.zeros(M, N, device="cuda", dtype=dtype) for _ in range(w)]
        outs = [torch.empty(M // w, N, device="cuda", dtype=dtype) for _ in range(w)]
        vc.run(lambda r, env: reduce_scatter_2d_op(xs[r], ctxs[r], outs[r]))        # odd number of calls before capture
        graphs = []
        for r in range(w):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=vc.streams[r]):
                reduce_scatter_2d_op(xs[r], ctxs[r], outs[r])
            graphs.append(g)
        Mr = M // w
        for it in range(6):
            for x in xs:
                x.copy_((torch.rand(M, N, device="cuda") * 2 - 1).to(dtype))
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()

            def step(r, env):
                if it == 2 and r == 0:
                    torch.cuda._sleep(10_000_000)
                if it == 3:
                    reduce_scatter_2d_op(xs[r], ctxs[r], outs[r])
                else:
                    graphs[r].replay()

            vc.run(step)
            ref = sum(x.float() for x in xs)
            for r in range(w):
                ctxs[r].check_error()
                assert torch.allclose(outs[r].float(), ref[r * Mr:(r + 1) * Mr], atol=3e-2, rtol=2e-2), (r, it)
    finally:
        vc.close()


def test_ring_reduce_helper():
    x = torch.randn(4 * 8, 16)
    out = torch.empty(8, 16)
    ring_reduce(x, out, begin_idx=1, num_splits=4)
    assert torch.allclose(out, x.view(4, 8, 16).sum(0), atol=1e-5)
