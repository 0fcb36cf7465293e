"""AG-GEMM and GEMM-RS on virtual ranks of one GPU vs fp32 references.

Scenarios follow the reference's python/triton_dist/test/amd/test_ag_gemm_intra_node.py and
test_gemm_rs_intra_node.py (--check loops with fresh random inputs, atol/rtol 1e-2) plus a straggler
rank (reference: kernels/nvidia/allgather_gemm.py:602-603 straggler_option).
"""
import pytest
import torch

from triton_distributed_gh_amd.ops.ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context
from triton_distributed_gh_amd.ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node
from triton_distributed_gh_amd.utils.testing import VirtualCluster, assert_close_elementwise

pytestmark = pytest.mark.gpu


def _ctas(w):
    return max(8, 200 // w)


@pytest.mark.parametrize("w,M,N,K,chunk,producer", [
    (2, 1024, 1024, 256, 256, "kernel"),
    (2, 2048, 1024, 512, 256, "kernel"),
    (4, 2048, 2048, 512, 256, "kernel"),
    (4, 256, 2048, 256, 256, "kernel"),       # 64 rows per rank: tiles straddle ranks
    (8, 4096, 2048, 256, 256, "kernel"),
    (8, 2048, 1024, 128, 128, "kernel"),
    (4, 256, 1536, 512, 256, "kernel"),       # 384 columns per rank: partial N tile
    (2, 1024, 1024, 512, 256, "copy_engine"),
    (4, 2048, 2048, 512, 256, "copy_engine"),
    (8, 4096, 2048, 256, 256, "copy_engine"),
    (4, 256, 2048, 256, 256, "copy_engine"),  # 64 rows per rank: tiles straddle ranks
    (4, 1024, 1024, 1024, 256, "kernel:split2"),
    (4, 1024, 2048, 2048, 128, "kernel:split4"),
    (4, 256, 1024, 512, 256, "kernel:split2"),   # straddling tiles + split-K
])
def test_ag_gemm_virtual_ranks(w, M, N, K, chunk, producer, monkeypatch):
    if ":split" in producer:
        producer, split = producer.split(":split")
        monkeypatch.setenv("TDIST_AG_SPLIT_K", split)
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ctxs = vc.collective(lambda r, env: create_ag_gemm_intra_node_context(
            M, N // w, K, torch.bfloat16, torch.bfloat16, r, w, M_PER_CHUNK=chunk, num_ctas=_ctas(w),
            producer=producer, env=env, ag_streams=[vc.side_streams[r]]))
        for it in range(3):
            a = [(torch.rand(M // w, K, device="cuda") * 2 - 1).bfloat16() for _ in range(w)]
            b = [(torch.rand(N // w, K, device="cuda") * 2 - 1).bfloat16() for _ in range(w)]
            torch.cuda.synchronize()

            def step(r, env):
                if it == 1 and r == w - 1:
                    torch.cuda._sleep(20_000_000)   # straggler: this rank's push and GEMM start late
                return ag_gemm_intra_node(a[r], b[r], False, ctxs[r])

            outs = vc.run(step)
            a_full = torch.cat(a).float()
            for r in range(w):
                ctxs[r].check_error()
                ref = a_full @ b[r].float().t()
                rel = (outs[r].float() - ref).abs().max().item() / ref.abs().max().item()
                assert rel < 1e-2, f"rank {r} iteration {it}: {rel}"
                assert_close_elementwise(outs[r], ref, what=f"ag_gemm rank {r} iteration {it}")
    finally:
        vc.close()


@pytest.mark.parametrize("w,M,N,K,chunk", [
    (2, 1024, 1024, 512, 256),
    (4, 2048, 1024, 1024, 256),
    (4, 4096, 2048, 512, 512),
    (8, 4096, 1024, 1024, 256),
    (8, 512, 768, 2048, 256),       # 64 rows per rank
    (4, 256, 512, 512, 256),        # K_local = 128: two K-steps only
])
def test_gemm_rs_virtual_ranks(w, M, N, K, chunk):
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ctxs = vc.collective(lambda r, env: create_gemm_rs_intra_node_context(
            M, N, torch.bfloat16, r, w, M_PER_CHUNK=chunk, num_ctas=_ctas(w), env=env))
        for it in range(3):
            a = [(torch.rand(M, K // w, device="cuda") * 2 - 1).bfloat16() for _ in range(w)]
            b = [(torch.rand(N, K // w, device="cuda") * 2 - 1).bfloat16() for _ in range(w)]
            torch.cuda.synchronize()

            def step(r, env):
                if it == 1 and r == 0:
                    torch.cuda._sleep(20_000_000)
                return gemm_rs_intra_node(a[r], b[r], ctxs[r])

            outs = vc.run(step)
            parts = [(a[r].float() @ b[r].float().t()).bfloat16().float() for r in range(w)]
            partial = sum(parts)
            mag = sum(p.abs() for p in parts)        # rounding of the 16-bit partials scales with this
            Mr = M // w
            for r in range(w):
                ctxs[r].check_error()
                ref = partial[r * Mr:(r + 1) * Mr]
                rel = (outs[r].float() - ref).abs().max().item() / ref.abs().max().item()
                assert rel < 1e-2, f"rank {r} iteration {it}: {rel}"
                assert_close_elementwise(outs[r], ref, what=f"gemm_rs rank {r} iteration {it}",
                                         magnitude=mag[r * Mr:(r + 1) * Mr])
    finally:
        vc.close()


@pytest.mark.parametrize("split", [1, 2, 3, 4])
def test_ag_gemm_split_k_single_rank(split, monkeypatch):
    """The fused kernel on one rank (no peers): split-K partial sums + last-arriver fix-up vs fp32."""
    monkeypatch.setenv("TDIST_FORCE_FUSED", "1")
    monkeypatch.setenv("TDIST_AG_SPLIT_K", str(split))
    vc = VirtualCluster(1, heap_bytes=1 << 26)
    try:
        M, N, K = 520, 776, 4096        # partial tiles in both directions
        ctx = create_ag_gemm_intra_node_context(M, N, K, torch.bfloat16, torch.bfloat16, 0, 1, env=vc.envs[0])
        for _ in range(2):
            a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
            b = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
            out = vc.run(lambda r, env: ag_gemm_intra_node(a, b, False, ctx))[0]
            ctx.check_error()
            ref = a.float() @ b.float().t()
            rel = (out.float() - ref).abs().max().item() / ref.abs().max().item()
            assert rel < 1e-2, rel
            assert_close_elementwise(out, ref, what=f"ag_gemm split-K {split}")
    finally:
        vc.close()


def test_fused_ops_on_a_grown_heap():
    """3 MiB segments: the GEMM-RS scatter buffers do not fit segment 0 next to the AG workspaces, so
    the kernels translate pointers that live in growth segments."""
    w, M, N, K = 2, 1024, 1024, 512
    vc = VirtualCluster(w, heap_bytes=3 << 20)
    try:
        ag = vc.collective(lambda r, env: create_ag_gemm_intra_node_context(M, N // w, K, torch.bfloat16, torch.bfloat16,
                                                                            r, w, num_ctas=_ctas(w), env=env))
        rs = vc.collective(lambda r, env: create_gemm_rs_intra_node_context(M, N, torch.bfloat16, r, w,
                                                                            num_ctas=_ctas(w), env=env))
        assert vc.envs[0].heap.ctx_struct().num_extra >= 1
        for it in range(3):
            a = [(torch.rand(M // w, K, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            b = [(torch.rand(N // w, K, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            a2 = [(torch.rand(M, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            b2 = [(torch.rand(N, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            torch.cuda.synchronize()
            o1 = vc.run(lambda r, env: ag_gemm_intra_node(a[r], b[r], False, ag[r]))
            o2 = vc.run(lambda r, env: gemm_rs_intra_node(a2[r], b2[r], rs[r]))
            a_full = torch.cat(a).float()
            part = sum((a2[r].float() @ b2[r].float().t()).bfloat16().float() for r in range(w))
            for r in range(w):
                ag[r].check_error()
                rs[r].check_error()
                ref = a_full @ b[r].float().t()
                assert (o1[r].float() - ref).abs().max().item() / ref.abs().max().item() < 1e-2
                ref2 = part[r * (M // w):(r + 1) * (M // w)]
                assert (o2[r].float() - ref2).abs().max().item() / ref2.abs().max().item() < 1e-2
    finally:
        vc.close()


@pytest.mark.parametrize("w,M,N,K", [(2, 1024, 1024, 512), (4, 1024, 2048, 1024), (2, 640, 1024, 512),
                                     (4, 1152, 1024, 512)])
def test_ag_gemm_gemm_rs_pair_replays_from_graph(w, M, N, K):
    """The TP MLP pair captured once per rank and replayed: both kernels number their calls on the
    device, so every replay (and an eager call in between) uses a fresh epoch and buffer parity.  The last
    two shapes give every rank 320 / 288 rows: 256-row tiles that straddle ranks (more rows than the
    low-latency forms take, so the 256x256-tile kernels run)."""
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ag = vc.collective(lambda r, env: create_ag_gemm_intra_node_context(
            M, N // w, K, torch.bfloat16, torch.bfloat16, r, w, num_ctas=_ctas(w), env=env,
            ag_streams=[vc.side_streams[r]]))
        rs = vc.collective(lambda r, env: create_gemm_rs_intra_node_context(M, K, torch.bfloat16, r, w,
                                                                            num_ctas=_ctas(w), env=env))
        x = [torch.zeros(M // w, K, device="cuda", dtype=torch.bfloat16) for _ in range(w)]
        w1 = [((torch.rand(N // w, K, device="cuda") * 2 - 1) * 0.05).bfloat16() for _ in range(w)]
        w2 = [((torch.rand(K, N // w, device="cuda") * 2 - 1) * 0.05).bfloat16() for _ in range(w)]
        outs = [None] * w

        def pair(r, env):
            h = ag_gemm_intra_node(x[r], w1[r], False, ag[r])          # [M, N/w]
            outs[r] = gemm_rs_intra_node(h, w2[r], rs[r])              # [M/w, K]
            return outs[r]

        def reference():
            xf = torch.cat(x).float()
            acc = 0
            for r in range(w):
                h = (xf @ w1[r].float().t()).bfloat16().float()
                acc = acc + (h @ w2[r].float().t()).bfloat16().float()
            return acc

        def fill():
            for t in x:
                t.copy_((torch.rand(M // w, K, device="cuda") * 2 - 1).bfloat16())
            torch.cuda.synchronize()

        def check(res):
            ref = reference()
            for r in range(w):
                ag[r].check_error()
                rs[r].check_error()
                got = res[r].float()
                want = ref[r * (M // w):(r + 1) * (M // w)]
                assert (got - want).abs().max().item() / want.abs().max().item() < 2e-2

        fill()
        check(vc.run(pair))                      # one eager call: an odd number of calls before capture
        graphs = []
        for r in range(w):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=vc.streams[r]):
                pair(r, vc.envs[r])
            graphs.append(g)
        static_outs = list(outs)
        for it in range(5):
            fill()
            for t in static_outs:
                t.zero_()
            torch.cuda.synchronize()
            if it == 2:
                check(vc.run(pair))
            else:
                vc.run(lambda r, env: graphs[r].replay())
                check(static_outs)
    finally:
        vc.close()
