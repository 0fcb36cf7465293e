"""Low-latency AG + GEMM and GEMM + RS (csrc/ops/tp_ll.hip) on virtual ranks: element-wise numerics at decode
shapes (M_per_rank 1 / 2 / 16 / 64), HIP-graph replay of the pair with fresh inputs."""
import pytest
import torch

from triton_distributed_gh_amd.ops.ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context
from triton_distributed_gh_amd.ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node
from triton_distributed_gh_amd.utils.testing import VirtualCluster

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _make(vc, w, M, N, K):
    ctas = max(8, 200 // w)
    ag = vc.collective(lambda r, env: create_ag_gemm_intra_node_context(M, N // w, K, BF, BF, r, w, num_ctas=ctas, env=env))
    rs = vc.collective(lambda r, env: create_gemm_rs_intra_node_context(M, K, BF, r, w, num_ctas=ctas, env=env))
    assert ag[0].ll is not None and rs[0].ll is not None
    return ag, rs


def _rand(*shape, scale=1.0):
    return ((torch.rand(*shape, device="cuda") * 2 - 1) * scale).to(BF)


def _check_pair(x, w1, w2, hs, outs, w, Mr, what):
    """h = AllGather(x) @ w1_r^T per rank (column shard), out = ReduceScatter(sum_r h_r @ w2_r^T)."""
    xf = torch.cat(x).float()
    acc, mag = 0, 0
    for r in range(w):
        href = xf @ w1[r].float().t()
        bad = (hs[r].float() - href).abs() > 1e-3 + 2e-2 * href.abs()
        assert not bad.any(), f"{what}: AG-GEMM rank {r}: {int(bad.sum())} elements off"
        part = (hs[r].float() @ w2[r].float().t()).to(BF).float()
        acc, mag = acc + part, mag + part.abs()
    for r in range(w):
        bad = (outs[r].float() - acc[r * Mr:(r + 1) * Mr]).abs() > 1e-3 + 2e-2 * mag[r * Mr:(r + 1) * Mr]
        assert not bad.any(), f"{what}: GEMM-RS rank {r}: {int(bad.sum())} elements off"


@pytest.mark.parametrize("w,Mr,N,K", [(2, 2, 1536, 512), (2, 2, 2048, 512), (2, 1, 1024, 512), (2, 64, 2048, 1024),
                                      (4, 16, 2048, 1024), (4, 1, 4096, 2048), (8, 16, 5120, 1024), (8, 2, 1024, 4096)])
def test_ll_pair_elementwise(w, Mr, N, K):
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        ag, rs = _make(vc, w, Mr * w, N, K)
        w1 = [_rand(N // w, K, scale=3.0 / K ** 0.5) for _ in range(w)]
        w2 = [_rand(K, N // w, scale=3.0 / N ** 0.5) for _ in range(w)]
        for it in range(4):
            x = [_rand(Mr, K) for _ in range(w)]
            torch.cuda.synchronize()
            hs = vc.run(lambda r, env: ag_gemm_intra_node(x[r], w1[r], False, ag[r]))
            outs = vc.run(lambda r, env: gemm_rs_intra_node(hs[r], w2[r], rs[r]))
            for r in range(w):
                ag[r].check_error()
                rs[r].check_error()
            _check_pair(x, w1, w2, hs, outs, w, Mr, f"iteration {it}")
    finally:
        vc.close()


@pytest.mark.parametrize("w,Mr", [(2, 1), (2, 16), (4, 16), (4, 64)])
def test_ll_pair_graph_replay(w, Mr):
    N, K = 2048, 1024
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        ag, rs = _make(vc, w, Mr * w, N, K)
        w1 = [_rand(N // w, K, scale=3.0 / K ** 0.5) for _ in range(w)]
        w2 = [_rand(K, N // w, scale=3.0 / N ** 0.5) for _ in range(w)]
        x = [torch.zeros(Mr, K, device="cuda", dtype=BF) for _ in range(w)]
        hs = [torch.zeros(Mr * w, N // w, device="cuda", dtype=BF) for _ in range(w)]
        outs = [torch.zeros(Mr, K, device="cuda", dtype=BF) for _ in range(w)]

        def step(r, env):
            hs[r].copy_(ag_gemm_intra_node(x[r], w1[r], False, ag[r]))
            outs[r].copy_(gemm_rs_intra_node(hs[r], w2[r], rs[r]))

        vc.run(step)
        graphs = []
        for r in range(w):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=vc.streams[r]):
                step(r, vc.envs[r])
            graphs.append(g)
        for it in range(7):
            for r in range(w):
                x[r].copy_(_rand(Mr, K))
                outs[r].zero_()
                hs[r].zero_()
            torch.cuda.synchronize()
            if it == 3:
                vc.run(step)
            else:
                vc.run(lambda r, env: graphs[r].replay())
            for r in range(w):
                ag[r].check_error()
                rs[r].check_error()
            _check_pair(x, w1, w2, hs, outs, w, Mr, f"replay {it}")
    finally:
        vc.close()
