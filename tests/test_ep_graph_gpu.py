"""Expert-parallel exchanges replayed from HIP graphs (virtual ranks on one GPU).

Scenarios: the DeepSeek decode all-to-all of the reference's tutorial (tutorials/04-deepseek-infer-all2all.py:
128 tokens x topk 8 per rank, hidden 7168 in fp8 e4m3 + one fp32 scale per 128 elements, 8 ranks; payload compared
bit-exact as python/triton_dist/test/nvidia/test_all_to_all.py does) and the normal-mode dispatch -> experts ->
combine round trip of python/triton_dist/test/nvidia/test_ep_a2a.py, each captured once per rank and replayed with
fresh inputs.  The call number / epoch lives in device memory, so a replay is a complete call."""
import pytest
import torch

from triton_distributed_gh_amd.ops.all_to_all import (all_to_all_post_process, create_all_to_all_context,
                                                      fast_all_to_all)
from triton_distributed_gh_amd.ops.ep_a2a import EPAll2AllLayer, dispatch_offsets
from triton_distributed_gh_amd.utils.testing import VirtualCluster

pytestmark = pytest.mark.gpu


def _capture(vc, fn):
    graphs = []
    for r in range(vc.world_size):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=vc.streams[r]):
            fn(r, vc.envs[r])
        graphs.append(g)
    return graphs


@pytest.mark.parametrize("w,max_m,hidden,dtype", [
    (8, 1024, 7168, torch.float8_e4m3fn),     # DeepSeek decode: 128 tokens x topk 8 per rank
    (4, 256, 1024, torch.bfloat16),
])
def test_fast_all_to_all_graph_replay(w, max_m, hidden, dtype):
    epr = 4
    E = w * epr
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        ctxs = vc.collective(lambda r, env: create_all_to_all_context(max_m, hidden, r, E, w, epr, dtype, env=env))
        send = [torch.zeros(max_m, hidden, device="cuda").to(dtype) for _ in range(w)]
        sscale = [torch.zeros(max_m, hidden // 128, device="cuda") for _ in range(w)]
        cum = [torch.zeros(E + 1, dtype=torch.int32, device="cuda") for _ in range(w)]
        res = [None] * w

        def step(r, env):
            splits, recv, rscale = fast_all_to_all(ctxs[r], send[r], cum[r], sscale[r])
            res[r] = all_to_all_post_process(ctxs[r], splits, recv, rscale, padded=True)

        def fresh():
            for r in range(w):
                counts = torch.randint(0, max_m // E + 1, (E,), device="cuda")
                cum[r][1:] = torch.cumsum(counts, 0)
                send[r].copy_((torch.randn(max_m, hidden, device="cuda") * 4).to(dtype))
                sscale[r].copy_(torch.rand(max_m, hidden // 128, device="cuda"))
            torch.cuda.synchronize()

        def check(what):
            for r in range(w):
                ctxs[r].check_error()
                toks, tscale, per_expert, total = res[r]
                rows = [send[s][int(cum[s][r * epr]):int(cum[s][(r + 1) * epr])] for s in range(w)]
                scs = [sscale[s][int(cum[s][r * epr]):int(cum[s][(r + 1) * epr])] for s in range(w)]
                want = torch.cat(rows, 0)
                n = int(total)
                assert n == want.shape[0], what
                assert torch.equal(toks[:n].view(torch.uint8), want.view(torch.uint8)), f"{what}: rank {r} rows"
                assert torch.equal(tscale[:n], torch.cat(scs, 0)), f"{what}: rank {r} scales"
                want_pe = sum((cum[s][r * epr + 1:(r + 1) * epr + 1] - cum[s][r * epr:(r + 1) * epr]) for s in range(w))
                assert torch.equal(per_expert, want_pe.int()), f"{what}: rank {r} tokens per expert"

        fresh()
        vc.run(step)                              # eager warm-up (also call 0 of the device sequence)
        torch.cuda.synchronize()
        check("eager")
        graphs = _capture(vc, step)
        captured = list(res)                      # the tensors the graphs write
        for it in range(5):                       # odd and even call numbers, both halves of the receive buffers
            fresh()
            if it == 3:
                vc.run(step)                      # an eager call between replays keeps the sequence
            else:
                res[:] = captured
                vc.run(lambda r, env: graphs[r].replay())
            torch.cuda.synchronize()
            check(f"iteration {it}")
    finally:
        vc.close()


def test_ep_counts_kernel_matches_reference_offsets():
    """The in-kernel offsets equal the tensor-op computation over the gathered [source, expert] table."""
    w, T, topk, hidden, epr = 4, 64, 4, 256, 3
    E = w * epr
    vc = VirtualCluster(w, heap_bytes=1 << 26)
    try:
        layers = vc.collective(lambda r, env: EPAll2AllLayer(None, T, hidden, topk, r, E, w, w, torch.bfloat16,
                                                             num_sm=4, capacity_factor=float(w), env=env))
        idx = [torch.stack([torch.randperm(E, device="cuda")[:topk] for _ in range(T)]).int() for _ in range(w)]
        torch.cuda.synchronize()
        vc.run(lambda r, env: layers[r].preprocess(idx[r]))
        torch.cuda.synchronize()
        table = torch.stack([torch.bincount(idx[s].reshape(-1).long(), minlength=E) for s in range(w)]).int()
        for r in range(w):
            layers[r].check_error()
            assert torch.equal(layers[r].table[r], table)
            dst_base, per_expert, num = dispatch_offsets(table, w, epr, r)
            assert torch.equal(layers[r].dst_base, dst_base)
            assert torch.equal(layers[r].recv_per_expert, per_expert)
            assert int(layers[r].num_recv_dev) == int(num)
    finally:
        vc.close()


@pytest.mark.parametrize("w,T,topk,hidden,epr,cap", [
    (8, 128, 8, 7168, 4, 2.0),        # DeepSeek-shaped: 128 tokens x topk 8 x 7168 per rank
    (4, 96, 2, 512, 2, 4.0),
])
def test_ep_dispatch_combine_graph_replay(w, T, topk, hidden, epr, cap):
    E = w * epr
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        layers = vc.collective(lambda r, env: EPAll2AllLayer(None, T, hidden, topk, r, E, w, w, torch.bfloat16,
                                                             num_sm=8, capacity_factor=cap, env=env))
        xs = [torch.zeros(T, hidden, device="cuda", dtype=torch.bfloat16) for _ in range(w)]
        idx = [torch.zeros(T, topk, device="cuda", dtype=torch.int32) for _ in range(w)]
        wts = [torch.zeros(T, topk, device="cuda") for _ in range(w)]
        outs = [torch.zeros(T, hidden, device="cuda", dtype=torch.bfloat16) for _ in range(w)]
        rows_id = torch.arange(layers[0].cap_rows, device="cuda")

        def step(r, env):
            L = layers[r]
            recv = L.dispatch(xs[r], idx[r], padded=True)                       # [cap_rows, hidden]
            ends = torch.cumsum(L.recv_per_expert, 0)
            e_local = torch.searchsorted(ends, rows_id, right=True).clamp_(max=epr - 1)
            factor = (r * epr + e_local + 1).to(torch.float32).unsqueeze(1)
            buf = L.expert_out_buffer()
            buf.copy_((recv.float() * factor).to(torch.bfloat16))               # "experts" write in place
            outs[r].copy_(L.combine(buf, wts[r]))

        def fresh():
            for r in range(w):
                xs[r].copy_(torch.randn(T, hidden, device="cuda").bfloat16())
                idx[r].copy_(torch.stack([torch.randperm(E, device="cuda")[:topk] for _ in range(T)]).int())
                wts[r].copy_(torch.rand(T, topk, device="cuda"))
                outs[r].zero_()
            torch.cuda.synchronize()

        def check(what):
            for r in range(w):
                layers[r].check_error()
                expect = sum(int(((idx[s] // epr) == r).sum()) for s in range(w))
                assert int(layers[r].num_recv_dev) == expect, what
                ref = torch.zeros(T, hidden, device="cuda")
                for k in range(topk):
                    e = idx[r][:, k].long()
                    ref += wts[r][:, k:k + 1] * (xs[r].float() * (e + 1).unsqueeze(1)).bfloat16().float()
                assert torch.allclose(outs[r].float(), ref, rtol=2e-2, atol=2e-2 * topk), f"{what}: rank {r}"

        fresh()
        vc.run(step)
        torch.cuda.synchronize()
        check("eager")
        graphs = _capture(vc, step)
        for it in range(4):
            fresh()
            vc.run(lambda r, env: graphs[r].replay())
            torch.cuda.synchronize()
            check(f"replay {it}")
    finally:
        vc.close()


def test_low_latency_dispatch_combine_round_trip_one_graph():
    """DeepSeek decode shape, 8 ranks: dispatch (fp8 rows + scales) -> compaction -> the return all-to-all, all in ONE
    graph per rank; the rows that come back are bit-identical to what was sent, in the original (expert-sorted)
    order (the gather / scatter pair of tutorials/04-deepseek-infer-all2all.py)."""
    w, epr, max_m, hidden = 8, 4, 1024, 7168
    E = w * epr
    vc = VirtualCluster(w, heap_bytes=1 << 29)
    try:
        mk = lambda r, env: create_all_to_all_context(max_m, hidden, r, E, w, epr, torch.float8_e4m3fn, env=env)  # noqa: E731
        disp, comb = vc.collective(mk), vc.collective(mk)
        send = [torch.zeros(max_m, hidden, device="cuda").to(torch.float8_e4m3fn) for _ in range(w)]
        sscale = [torch.zeros(max_m, hidden // 128, device="cuda") for _ in range(w)]
        cum = [torch.zeros(E + 1, dtype=torch.int32, device="cuda") for _ in range(w)]
        back_cum = [torch.zeros(E + 1, dtype=torch.int32, device="cuda") for _ in range(w)]
        back = [None] * w

        def step(r, env):
            splits, recv, rscale = fast_all_to_all(disp[r], send[r], cum[r], sscale[r])
            toks, tscale, _, _ = all_to_all_post_process(disp[r], splits, recv, rscale, padded=True)
            # the compacted rows are ordered by source rank = by destination of the way back; "expert" slot j of
            # destination d carries what d sent for local expert j
            back_cum[r][1:] = torch.cumsum(splits.reshape(-1), 0)
            back[r] = fast_all_to_all(comb[r], toks[:max_m], back_cum[r], tscale[:max_m])

        def fresh():
            for r in range(w):
                # every destination rank gets at most max_m / w rows from each source, so the way back fits max_m
                counts = torch.randint(0, max_m // E + 1, (E,), device="cuda")
                cum[r][1:] = torch.cumsum(counts, 0)
                send[r].copy_((torch.randn(max_m, hidden, device="cuda") * 4).to(torch.float8_e4m3fn))
                sscale[r].copy_(torch.rand(max_m, hidden // 128, device="cuda"))
            torch.cuda.synchronize()

        def check(what):
            for r in range(w):
                disp[r].check_error()
                comb[r].check_error()
                _, rows, scales = back[r]
                for d in range(w):
                    st, ed = int(cum[r][d * epr]), int(cum[r][(d + 1) * epr])
                    assert torch.equal(rows[d, : ed - st].view(torch.uint8), send[r][st:ed].view(torch.uint8)), f"{what}: {r}<-{d}"
                    assert torch.equal(scales[d, : ed - st], sscale[r][st:ed]), f"{what}: scales {r}<-{d}"

        fresh()
        vc.run(step)
        torch.cuda.synchronize()
        check("eager")
        graphs = _capture(vc, step)
        for it in range(3):
            fresh()
            vc.run(lambda r, env: graphs[r].replay())
            torch.cuda.synchronize()
            check(f"replay {it}")
    finally:
        vc.close()
