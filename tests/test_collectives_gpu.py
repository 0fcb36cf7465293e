"""AllReduce family, GEMM+AR, low-latency all-gather, low-latency all-to-all and EP dispatch/combine
on virtual ranks of one GPU.  Goldens are plain PyTorch (scenarios of the reference's
test_allreduce.py, test_gemm_ar.py, test_fast_allgather.py, test_all_to_all.py, test_ep_a2a.py;
a2a / dispatch payloads are compared bit-exact as in test_ep_a2a.py:47-53)."""
import pytest
import torch

from triton_distributed_gh_amd.ops.all_to_all import (all_to_all_post_process, create_all_to_all_context,
                                                      dequantize_fp8_groupwise, fast_all_to_all,
                                                      quantize_fp8_groupwise)
from triton_distributed_gh_amd.ops.allgather import create_fast_allgather_context, fast_allgather
from triton_distributed_gh_amd.ops.allreduce import AllReduceMethod, all_reduce, create_allreduce_ctx
from triton_distributed_gh_amd.ops.ep_a2a import EPAll2AllLayer
from triton_distributed_gh_amd.ops.gemm_ar import create_gemm_ar_context, gemm_allreduce_op
from triton_distributed_gh_amd.utils.testing import VirtualCluster, assert_close_elementwise

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w", [2, 4, 8])
@pytest.mark.parametrize("method", [AllReduceMethod.OneShot, AllReduceMethod.TwoShot, AllReduceMethod.DoubleTree])
@pytest.mark.parametrize("dtype,n", [(torch.bfloat16, 8192 * 3 + 8), (torch.float32, 100_003), (torch.float16, 1 << 20)])
def test_all_reduce(w, method, dtype, n):
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ctxs = vc.collective(lambda r, env: create_allreduce_ctx(8 << 20, r, w, env=env))
        for it in range(3):
            xs = [(torch.rand(n, device="cuda") - 0.5).to(dtype) for _ in range(w)]
            torch.cuda.synchronize()
            outs = vc.run(lambda r, env: all_reduce(xs[r], method, ctxs[r],
                                                    straggler_option=(1, 5_000_000) if it == 1 else None))
            ref = sum(x.float() for x in xs)
            for r in range(w):
                ctxs[r].check_error()
                assert torch.allclose(outs[r].float(), ref, atol=2e-2 * w, rtol=2e-2)
                assert torch.equal(outs[r], outs[0]), "ranks must agree bitwise"
    finally:
        vc.close()


@pytest.mark.parametrize("w", [2, 4])
@pytest.mark.parametrize("method", [AllReduceMethod.OneShot, AllReduceMethod.TwoShot])
def test_all_reduce_graph_replay(w, method):
    """The call number lives on the device, so a captured all-reduce picks a fresh epoch and buffer
    parity on every replay: replays, eager calls in between and two launches per graph all agree."""
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ctxs = vc.collective(lambda r, env: create_allreduce_ctx(8 << 20, r, w, env=env))
        n = 8192 * 3 + 8
        xs = [torch.zeros(n, device="cuda", dtype=torch.bfloat16) for _ in range(w)]
        mid = [torch.empty_like(x) for x in xs]
        outs = [torch.empty_like(x) for x in xs]

        def fill():
            for x in xs:
                x.copy_((torch.rand(n, device="cuda") - 0.5).bfloat16())
            torch.cuda.synchronize()
            first = sum(x.float() for x in xs).bfloat16()
            return first, (first.float() * w).bfloat16()

        def check(first, second):
            for r in range(w):
                ctxs[r].check_error()
                assert torch.allclose(mid[r].float(), first.float(), atol=2e-2 * w, rtol=2e-2)
                assert torch.allclose(outs[r].float(), second.float(), atol=4e-2 * w * w, rtol=4e-2)
                assert torch.equal(outs[r], outs[0])

        def eager(r, env):
            all_reduce(xs[r], method, ctxs[r], output=mid[r])
            all_reduce(mid[r], method, ctxs[r], output=outs[r])

        fill()
        vc.run(lambda r, env: all_reduce(xs[r], method, ctxs[r], output=mid[r]))   # odd number of calls before capture
        graphs = []
        for r in range(w):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=vc.streams[r]):
                eager(r, vc.envs[r])
            graphs.append(g)
        for it in range(6):
            first, second = fill()
            for t in mid + outs:
                t.zero_()
            torch.cuda.synchronize()
            if it == 3:
                vc.run(eager)                  # an eager call between replays keeps the sequence
            else:
                vc.run(lambda r, env: graphs[r].replay())
            check(first, second)
    finally:
        vc.close()


@pytest.mark.parametrize("w,M,N,K", [(2, 512, 1024, 512), (4, 128, 2048, 1024), (8, 300, 1280, 2048)])
def test_gemm_allreduce(w, M, N, K):
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ctxs = vc.collective(lambda r, env: create_gemm_ar_context(None, r, w, w, M, N, torch.bfloat16,
                                                                   num_ctas=max(8, 200 // w), env=env))
        for it in range(3):
            a = [(torch.rand(M, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            b = [(torch.rand(N, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            torch.cuda.synchronize()
            outs = vc.run(lambda r, env: gemm_allreduce_op(ctxs[r], a[r], b[r]))
            parts = [(a[r].float() @ b[r].float().t()).bfloat16().float() for r in range(w)]
            ref = sum(parts)
            mag = sum(p.abs() for p in parts)        # rounding of the 16-bit partials scales with this
            for r in range(w):
                ctxs[r].check_error()
                rel = (outs[r].float() - ref).abs().max().item() / ref.abs().max().item()
                assert rel < 1e-2
                assert_close_elementwise(outs[r], ref, what=f"gemm_allreduce rank {r}", magnitude=mag)
                assert torch.equal(outs[r], outs[0])
    finally:
        vc.close()


@pytest.mark.parametrize("w", [2, 4, 8])
@pytest.mark.parametrize("mode", ["pull", "push_2d", "push_2d_ll", "push_numa_2d_ll"])
def test_fast_allgather(w, mode):
    vc = VirtualCluster(w, heap_bytes=1 << 26)
    try:
        shard = (96, 130)   # bf16 -> 24960 bytes per rank
        ctxs = vc.collective(lambda r, env: create_fast_allgather_context(r, max_buffer_size=shard[0] * shard[1] * 2, env=env))
        bufs = vc.collective(lambda r, env: env.heap.create_tensor([w, *shard], torch.bfloat16))
        for it in range(4):
            vals = [torch.randn(shard, device="cuda").bfloat16() for _ in range(w)]
            for r in range(w):
                bufs[r][r].copy_(vals[r])
            torch.cuda.synchronize()
            vc.run(lambda r, env: fast_allgather(bufs[r], ctxs[r], mode=mode))
            for r in range(w):
                ctxs[r].check_error()
                for s in range(w):
                    assert torch.equal(bufs[r][s], vals[s])
    finally:
        vc.close()


@pytest.mark.parametrize("w", [2, 4])
@pytest.mark.parametrize("mode", ["pull", "push_2d", "push_2d_ll"])
def test_fast_allgather_graph_replay(w, mode):
    """Every form of the all-gather numbers its calls on the device: captured once per rank, replayed with fresh
    shards, an eager call in between."""
    vc = VirtualCluster(w, heap_bytes=1 << 26)
    try:
        shard = (300, 520)   # bf16 -> 312000 bytes per rank: two slices per peer in the bulk forms
        ctxs = vc.collective(lambda r, env: create_fast_allgather_context(r, max_buffer_size=shard[0] * shard[1] * 2, env=env))
        bufs = vc.collective(lambda r, env: env.heap.create_tensor([w, *shard], torch.bfloat16))
        vc.run(lambda r, env: fast_allgather(bufs[r], ctxs[r], mode=mode))        # odd number of calls before capture
        graphs = []
        for r in range(w):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=vc.streams[r]):
                fast_allgather(bufs[r], ctxs[r], mode=mode)
            graphs.append(g)
        for it in range(6):
            vals = [torch.randn(shard, device="cuda").bfloat16() for _ in range(w)]
            for r in range(w):
                bufs[r].zero_()
                bufs[r][r].copy_(vals[r])
            torch.cuda.synchronize()
            if it == 3:
                vc.run(lambda r, env: fast_allgather(bufs[r], ctxs[r], mode=mode))
            else:
                vc.run(lambda r, env: graphs[r].replay())
            for r in range(w):
                ctxs[r].check_error()
                for s_ in range(w):
                    assert torch.equal(bufs[r][s_], vals[s_])
    finally:
        vc.close()


@pytest.mark.parametrize("w,with_scale,max_m,hidden", [
    (2, False, 256, 1024), (4, True, 256, 1024), (8, True, 256, 1024),
    (4, True, 1024, 2048),       # ~1 MiB per destination: 8 workgroups per destination
    (8, False, 1024, 7168),      # DeepSeek-shaped rows (128 tokens x topk 8 per rank)
])
def test_fast_all_to_all(w, with_scale, max_m, hidden):
    epr = 4
    E = w * epr
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        ctxs = vc.collective(lambda r, env: create_all_to_all_context(max_m, hidden, r, E, w, epr, torch.bfloat16, env=env))
        for it in range(3):
            sends, cums, scales = [], [], []
            for r in range(w):
                counts = torch.randint(0, max_m // E, (E,), device="cuda")
                cum = torch.zeros(E + 1, dtype=torch.int32, device="cuda")
                cum[1:] = torch.cumsum(counts, 0)
                n = int(cum[-1])
                sends.append(torch.randn(n, hidden, device="cuda").bfloat16())
                scales.append(torch.rand(n, hidden // 128, device="cuda"))
                cums.append(cum)
            torch.cuda.synchronize()
            outs = vc.run(lambda r, env: fast_all_to_all(ctxs[r], sends[r], cums[r], scales[r] if with_scale else None))
            for r in range(w):
                ctxs[r].check_error()
                splits, recv, rscale = outs[r]
                for s in range(w):
                    st, ed = int(cums[s][r * epr]), int(cums[s][(r + 1) * epr])
                    assert torch.equal(recv[s, : ed - st], sends[s][st:ed])
                    assert torch.equal(splits[s].cpu(), (cums[s][r * epr + 1:(r + 1) * epr + 1] - cums[s][r * epr:(r + 1) * epr]).cpu())
                    if with_scale:
                        assert torch.equal(rscale[s, : ed - st], scales[s][st:ed])
                toks, _, per_exp = all_to_all_post_process(ctxs[r], splits, recv, rscale)
                assert toks.shape[0] == int(per_exp.sum())
    finally:
        vc.close()


def test_fp8_group_quant_roundtrip():
    x = (torch.randn(64, 1024, device="cuda") * 3).bfloat16()
    q, s = quantize_fp8_groupwise(x)
    assert q.dtype == torch.float8_e4m3fn and s.shape == (64, 8)
    ref_scale = x.float().view(64, 8, 128).abs().amax(-1) / 448.0
    assert torch.allclose(s, ref_scale, rtol=1e-6, atol=1e-12)
    back = dequantize_fp8_groupwise(q, s)
    assert (back.float() - x.float()).abs().max().item() <= x.float().abs().max().item() * 0.07


@pytest.mark.parametrize("w,T,topk", [(2, 64, 2), (4, 200, 4), (8, 128, 8)])
def test_ep_dispatch_combine(w, T, topk):
    hidden, epr = 512, 2
    E = w * epr
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        layers = vc.collective(lambda r, env: EPAll2AllLayer(None, T, hidden, topk, r, E, w, w, torch.bfloat16,
                                                             num_sm=4, capacity_factor=float(w), env=env))
        for it in range(2):
            xs = [torch.randn(T, hidden, device="cuda").bfloat16() for _ in range(w)]
            idx = [torch.stack([torch.randperm(E, device="cuda")[:topk] for _ in range(T)]).int() for _ in range(w)]
            wts = [torch.rand(T, topk, device="cuda") for _ in range(w)]
            torch.cuda.synchronize()
            recvs = _dispatch_all(vc, layers, xs, idx)
            # every received row must be one of the source tokens routed to a local expert
            for r in range(w):
                layers[r].check_error()
                n = layers[r].num_recv
                expect = sum(int(((idx[s] // epr) == r).sum()) for s in range(w))
                assert n == expect
            # "expert compute" = multiply by (global expert id + 1)
            outs_e = []
            for r in range(w):
                per = layers[r].recv_per_expert.tolist()
                rows = recvs[r].clone()
                o = 0
                for e_local, cnt in enumerate(per):
                    rows[o:o + cnt] = (rows[o:o + cnt].float() * (r * epr + e_local + 1)).bfloat16()
                    o += cnt
                outs_e.append(rows)
            torch.cuda.synchronize()
            outs = vc.run(lambda r, env: layers[r].combine(outs_e[r], wts[r]))
            for r in range(w):
                layers[r].check_error()
                ref = torch.zeros(T, hidden, device="cuda")
                for k in range(topk):
                    e = idx[r][:, k].long()
                    ref += wts[r][:, k:k + 1] * (xs[r].float() * (e + 1).unsqueeze(1)).bfloat16().float()
                assert torch.allclose(outs[r].float(), ref, rtol=2e-2, atol=2e-2 * topk)
    finally:
        vc.close()


def test_ep_two_dispatches_in_a_row():
    """A dispatch that is not followed by a combine: the next dispatch is complete on its own (cursors, counters and
    flags are re-armed by the kernels), the combine belongs to the second, and a second combine is refused."""
    w, T, topk, hidden, epr = 4, 96, 2, 256, 2
    E = w * epr
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        layers = vc.collective(lambda r, env: EPAll2AllLayer(None, T, hidden, topk, r, E, w, w, torch.bfloat16,
                                                             num_sm=4, capacity_factor=float(w), env=env))
        def fresh():
            xs = [torch.randn(T, hidden, device="cuda").bfloat16() for _ in range(w)]
            idx = [torch.stack([torch.randperm(E, device="cuda")[:topk] for _ in range(T)]).int() for _ in range(w)]
            torch.cuda.synchronize()
            return xs, idx

        xs1, idx1 = fresh()
        _dispatch_all(vc, layers, xs1, idx1)
        xs2, idx2 = fresh()
        second = _dispatch_all(vc, layers, xs2, idx2)
        for r in range(w):
            layers[r].check_error()
        outs = vc.run(lambda r, env: layers[r].combine(second[r].clone()))
        for r in range(w):
            layers[r].check_error()
            assert torch.allclose(outs[r].float(), xs2[r].float() * topk, rtol=2e-2, atol=2e-2 * topk)
        with pytest.raises(RuntimeError, match="one combine per dispatch"):
            layers[0].combine(second[0].clone())
    finally:
        vc.close()


def _dispatch_all(vc, layers, xs, idx):
    """dispatch() reads the receive count on the host after the counts exchange, which blocks until
    every rank has launched its exchange kernel - so launch the exchanges of all ranks first."""
    import threading

    outs = [None] * vc.world_size
    errs = []

    def work(r):
        try:
            with torch.cuda.stream(vc.streams[r]):
                outs[r] = layers[r].dispatch(xs[r], idx[r])
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(r,)) for r in range(vc.world_size)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    torch.cuda.synchronize()
    return outs
