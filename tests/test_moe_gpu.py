"""MoE kernels vs PyTorch goldens (masked per-expert matmul loops as in the reference's
test_ag_moe.py:39-55 and test_moe_reduce_rs.py:84-103; index utilities as in test_moe_utils.py)."""
import pytest
import torch

from triton_distributed_gh_amd.ops.moe import (_align_cpu, ag_group_gemm, create_ag_group_gemm_context,
                                               create_moe_rs_context, moe_align_block_size, run_moe_reduce_rs)
from triton_distributed_gh_amd.utils.testing import VirtualCluster, assert_close_elementwise

pytestmark = pytest.mark.gpu


def _rand_topk(ntokens, E, topk):
    return torch.stack([torch.randperm(E, device="cuda")[:topk] for _ in range(ntokens)]).int()


@pytest.mark.parametrize("ntokens,E,topk,w,rank", [(64, 8, 2, 1, 0), (512, 8, 2, 4, 1), (300 * 8, 32, 4, 8, 5)])
def test_moe_align_matches_cpu_oracle(ntokens, E, topk, w, rank):
    ids = _rand_topk(ntokens, E, topk)
    tpr = ntokens // w
    got = moe_align_block_size(ids, E, w, rank, tpr)
    cap = got.sorted_ids.numel()
    ref = _align_cpu(ids.cpu(), E, w, rank, tpr, cap)
    nb = int(ref.meta[0])
    assert torch.equal(got.meta.cpu(), ref.meta)
    assert torch.equal(got.expert_of_block.cpu()[:nb], ref.expert_of_block[:nb])
    assert torch.equal(got.seg_range.cpu()[:nb], ref.seg_range[:nb])
    assert torch.equal(got.order.cpu()[:nb], ref.order[:nb])
    # inside an (expert, source-rank) segment the order is arbitrary: compare as sets per segment
    g, r = got.sorted_ids.cpu(), ref.sorted_ids
    assert torch.equal(g < 0, r < 0)
    flat_e = ids.cpu().reshape(-1).long()
    src = (torch.arange(ntokens * topk) // topk // tpr).clamp(max=w - 1)
    key_g = torch.where(g >= 0, flat_e[g.clamp(min=0).long()] * w + src[g.clamp(min=0).long()], torch.tensor(-1))
    key_r = torch.where(r >= 0, flat_e[r.clamp(min=0).long()] * w + src[r.clamp(min=0).long()], torch.tensor(-1))
    assert torch.equal(key_g, key_r)
    assert torch.equal(torch.sort(g).values, torch.sort(r).values)


@pytest.mark.parametrize("w,ntokens,N,K,E,topk", [(1, 256, 512, 256, 4, 2), (2, 512, 512, 256, 8, 2),
                                                  (4, 1024, 1024, 512, 8, 2), (8, 1024, 2048, 256, 16, 4)])
def test_ag_group_gemm(w, ntokens, N, K, E, topk):
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ctxs = vc.collective(lambda r, env: create_ag_group_gemm_context(
            ntokens, N // w, K, E, topk, torch.bfloat16, r, w, num_ctas=max(8, 200 // w),
            ag_streams=[vc.side_streams[r]], env=env))
        for it in range(2):
            a = [(torch.rand(ntokens // w, K, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            b = [(torch.rand(E, N // w, K, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            ids = _rand_topk(ntokens, E, topk)
            torch.cuda.synchronize()
            outs = vc.run(lambda r, env: ag_group_gemm(a[r], b[r], ctxs[r], ids))
            a_full = torch.cat(a).float()
            for r in range(w):
                ctxs[r].check_error()
                ref = torch.empty(ntokens * topk, N // w, device="cuda")
                flat = ids.reshape(-1).long()
                rows = a_full.repeat_interleave(topk, dim=0)
                for e in range(E):
                    m = flat == e
                    ref[m] = rows[m] @ b[r][e].float().t()
                rel = (outs[r].float() - ref).abs().max().item() / ref.abs().max().item()
                assert rel < 1e-2, f"rank {r}: {rel}"
                assert_close_elementwise(outs[r], ref, what=f"ag_group_gemm rank {r}")
    finally:
        vc.close()


@pytest.mark.parametrize("w,ntokens,hidden,K,E,topk", [(1, 128, 512, 256, 4, 2), (2, 256, 512, 512, 8, 2),
                                                       (4, 512, 1024, 1024, 8, 2), (8, 512, 768, 2048, 16, 4),
                                                       (4, 88, 520, 512, 8, 2)])   # 22 tokens per rank: partial token group
@pytest.mark.parametrize("n_chunks", [1, 3])
def test_moe_reduce_rs(w, ntokens, hidden, K, E, topk, n_chunks):
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ctxs = vc.collective(lambda r, env: create_moe_rs_context(r, w, w, ntokens, hidden, E, topk, torch.bfloat16,
                                                                  num_ctas=max(8, 200 // w), env=env,
                                                                  rs_stream=vc.side_streams[r]))
        for it in range(2):
            x = [(torch.rand(ntokens * topk, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            wt = [(torch.rand(E, hidden, K // w, device="cuda") - 0.5).bfloat16() for _ in range(w)]
            ids = _rand_topk(ntokens, E, topk)
            rw = torch.rand(ntokens, topk, device="cuda")
            torch.cuda.synchronize()
            # n_chunks = 3: bands of whole 256-column tiles (the last one ragged at hidden = 520 / 768), each
            # reduced on a second stream while the next band multiplies
            outs = vc.run(lambda r, env: run_moe_reduce_rs(x[r], wt[r], ids, rw, ctxs[r], n_chunks=n_chunks))
            flat = ids.reshape(-1).long()
            total = torch.zeros(ntokens, hidden, device="cuda")
            mag = torch.zeros(ntokens, hidden, device="cuda")      # sum of |16-bit partials| per output element
            for r in range(w):
                part = torch.empty(ntokens * topk, hidden, device="cuda")
                for e in range(E):
                    m = flat == e
                    part[m] = x[r][m].float() @ wt[r][e].float().t()
                part = (part * rw.reshape(-1, 1)).bfloat16().float()
                mag += part.abs().view(ntokens, topk, hidden).sum(1)
                total += part.view(ntokens, topk, hidden).sum(1).bfloat16().float() if w > 1 else part.view(ntokens, topk, hidden).sum(1)
            tpr = ntokens // w
            for r in range(w):
                ctxs[r].check_error()
                ref = total[r * tpr:(r + 1) * tpr]
                rel = (outs[r].float() - ref).abs().max().item() / ref.abs().max().item()
                assert rel < 2e-2, f"rank {r}: {rel}"
                assert_close_elementwise(outs[r], ref, rtol=1e-2, atol=1e-2, what=f"moe_reduce_rs rank {r}",
                                         magnitude=mag[r * tpr:(r + 1) * tpr])
    finally:
        vc.close()


@pytest.mark.parametrize("n,H,I,E,topk", [(1, 2048, 768, 128, 8), (4, 2048, 768, 128, 8), (16, 512, 256, 16, 2),
                                          (3, 4096, 1536, 32, 8), (2, 1024, 96, 8, 3), (5, 512, 256, 16, 10)])
def test_moe_decode_kernels_match_fp32_reference(n, H, I, E, topk):
    """The two decode launches (per-pair decode GEMM with SiLU; down projection with the routing weights and the
    top-k sum folded into the reduction over waves) against fp32 PyTorch, element by element."""
    from triton_distributed_gh_amd.ops.moe import moe_decode_fwd, moe_decode_supported

    torch.manual_seed(n * 100 + topk)
    dev, bf = "cuda", torch.bfloat16
    x = (torch.rand(n, H, device=dev) * 2 - 1).to(bf)
    gate_up = ((torch.rand(E, 2 * I, H, device=dev) * 2 - 1) * 0.05).to(bf)
    down = ((torch.rand(E, H, I, device=dev) * 2 - 1) * 0.05).to(bf)
    ids = torch.stack([torch.randperm(E, device=dev)[:topk] for _ in range(n)])
    wts = torch.rand(n, topk, device=dev) + 0.1
    wts = wts / wts.sum(-1, keepdim=True)
    assert moe_decode_supported(x, gate_up, down, topk)
    ref = torch.zeros(n, H, device=dev)
    for t in range(n):
        for k in range(topk):
            e = int(ids[t, k])
            gu = gate_up[e].float() @ x[t].float()
            h = (torch.nn.functional.silu(gu[:I].to(bf).float()) * gu[I:].to(bf).float()).to(bf).float()
            ref[t] += wts[t, k] * (down[e].float() @ h)
    for form in (1, 0):                    # both kernels: vector-ALU form, MFMA form
        out = moe_decode_fwd(x, ids, wts, gate_up, down, down_form=form, gate_up_form=form)
        torch.testing.assert_close(out.float(), ref, atol=2e-2 * ref.abs().max().item(), rtol=2e-2)


@pytest.mark.parametrize("n,H,I,E,topk", [(12, 2048, 768, 128, 8), (17, 2048, 768, 128, 8), (40, 2048, 768, 128, 8), (200, 1024, 96, 64, 8),
                                          (100, 512, 2304, 16, 4), (64, 512, 256, 8, 2), (33, 1056, 160, 12, 5)])
def test_moe_block_grouped_by_expert_in_blocks_of_sixteen(n, H, I, E, topk, monkeypatch):
    """17 .. 256 tokens: pairs sorted by expert into sixteen-row blocks, gate_up + SiLU and down per block, top-k sum
    (``moe_decode_fwd`` above ``MOE_DECODE_MAX_TOKENS`` tokens) against fp32 PyTorch; experts with no token, with a
    partial block and with several blocks all occur.  An expert id out of range contributes nothing."""
    from triton_distributed_gh_amd.ops import moe as moe_ops
    from triton_distributed_gh_amd.ops.moe import moe_decode_fwd, moe_decode_supported

    torch.manual_seed(n * 10 + topk)
    dev, bf = "cuda", torch.bfloat16
    x = (torch.rand(n, H, device=dev) * 2 - 1).to(bf)
    gate_up = ((torch.rand(E, 2 * I, H, device=dev) * 2 - 1) * 0.05).to(bf)
    down = ((torch.rand(E, H, I, device=dev) * 2 - 1) * 0.05).to(bf)
    # skewed routing: low expert indices are chosen more often, the last expert never
    pick = torch.rand(n, E - 1, device=dev) * torch.linspace(1.0, 0.2, E - 1, device=dev)
    ids = torch.topk(pick, topk, dim=1).indices.int()
    wts = torch.rand(n, topk, device=dev) + 0.1
    wts = wts / wts.sum(-1, keepdim=True)
    assert moe_decode_supported(x, gate_up, down, topk) and moe_ops._group16_takes(n, topk, E, I, H)
    xf = x.float()
    ref = torch.zeros(n, H, device=dev)
    for e in range(E):
        tok, k = torch.where(ids == e)
        if tok.numel() == 0:
            continue
        gu = xf[tok] @ gate_up[e].float().t()
        h = (torch.nn.functional.silu(gu[:, :I].to(bf).float()) * gu[:, I:].to(bf).float()).to(bf).float()
        ref.index_add_(0, tok, (h @ down[e].float().t()) * wts[tok, k, None])
    out = moe_decode_fwd(x, ids, wts, gate_up, down)
    torch.testing.assert_close(out.float(), ref, atol=2e-2 * ref.abs().max().item(), rtol=2e-2)
    for rows in (16, 32):                  # both block heights, whatever the rule picked above
        monkeypatch.setattr(moe_ops, "MOE_GROUP_BLOCK_ROWS", rows)
        o2 = moe_decode_fwd(x, ids, wts, gate_up, down)
        torch.testing.assert_close(o2.float(), ref, atol=2e-2 * ref.abs().max().item(), rtol=2e-2)
    monkeypatch.setattr(moe_ops, "MOE_GROUP_BLOCK_ROWS", 0)
    bad = ids.clone()
    bad[1, 0] = 0x7fffffff
    bad[2, 1] = -3
    out2 = moe_decode_fwd(x, bad, wts, gate_up, down)
    torch.cuda.synchronize()
    assert torch.isfinite(out2.float()).all() and torch.equal(out2[0], out[0]) and torch.equal(out2[3:], out[3:])


@pytest.mark.parametrize("block_rows,wn", [(32, 1), (32, 2), (64, 1), (64, 2), (128, 1), (128, 2)])
@pytest.mark.parametrize("n,H,I,E,topk", [(600, 2048, 768, 128, 8), (300, 512, 256, 8, 2), (200, 1024, 128, 12, 5)])
def test_moe_block_grouped_by_expert_in_blocks_of_64_and_128(n, H, I, E, topk, block_rows, wn):
    """Hundreds of tokens on one rank: pairs sorted by expert into 32- / 64- / 128-row blocks, gathered activations, the mid-M
    GEMM body per block for gate_up and down, weighted top-k sum through the inverse map (``moe_group_mid_fwd``)
    against fp32 PyTorch; skewed routing, one expert without tokens, ids out of range contribute nothing."""
    from triton_distributed_gh_amd.ops.moe import moe_group_mid_fwd, moe_group_mid_supported

    torch.manual_seed(n + topk)
    dev, bf = "cuda", torch.bfloat16
    x = (torch.rand(n, H, device=dev) * 2 - 1).to(bf)
    gate_up = ((torch.rand(E, 2 * I, H, device=dev) * 2 - 1) * 0.05).to(bf)
    down = ((torch.rand(E, H, I, device=dev) * 2 - 1) * 0.05).to(bf)
    pick = torch.rand(n, E - 1, device=dev) * torch.linspace(1.0, 0.2, E - 1, device=dev)
    ids = torch.topk(pick, topk, dim=1).indices.int()
    wts = torch.rand(n, topk, device=dev) + 0.1
    wts = wts / wts.sum(-1, keepdim=True)
    assert moe_group_mid_supported(x, gate_up, down, topk)
    xf = x.float()
    ref = torch.zeros(n, H, device=dev)
    for e in range(E):
        tok, k = torch.where(ids == e)
        if tok.numel() == 0:
            continue
        gu = (xf[tok] @ gate_up[e].float().t()).to(bf).float()
        h = (torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]).to(bf).float()
        ref.index_add_(0, tok, (h @ down[e].float().t()).to(bf).float() * wts[tok, k, None])
    out = moe_group_mid_fwd(x, ids, wts, gate_up, down, block_rows=block_rows, wn=wn)
    torch.testing.assert_close(out.float(), ref, atol=2e-2 * ref.abs().max().item(), rtol=2e-2)
    bad = ids.clone()
    bad[1, 0] = 0x7fffffff
    bad[2, 1] = -3
    out2 = moe_group_mid_fwd(x, bad, wts, gate_up, down, block_rows=block_rows, wn=wn)
    torch.cuda.synchronize()
    assert torch.isfinite(out2.float()).all() and torch.equal(out2[0], out[0]) and torch.equal(out2[3:], out[3:])


@pytest.mark.parametrize("n,E,k,renorm", [(1, 128, 8, True), (7, 128, 8, False), (33, 8, 2, True), (4, 1000, 10, True),
                                          (300, 64, 6, True), (2, 4, 4, True)])
def test_router_epilogue_matches_torch_topk(n, E, k, renorm):
    """Fused softmax + top-k + renormalisation against the torch lines it replaces, including rows full of ties
    (bf16 logits collide often): same expert SET per token, same weight per chosen expert."""
    from triton_distributed_gh_amd.ops.moe import moe_route, moe_route_supported

    torch.manual_seed(n + E + k)
    logits = (torch.randn(n, E, device="cuda") * 2).to(torch.bfloat16)
    logits[0, : E // 2] = logits[0, 0]                                 # heavy ties in one row
    if n > 1:
        logits[1] = 0.5                                                # a row of equal logits: lowest indices win
    assert moe_route_supported(logits, k)
    wts, ids = moe_route(logits, k, renorm)
    probs = torch.softmax(logits, dim=1, dtype=torch.float)
    rw, rid = torch.topk(probs, k, dim=-1)
    if renorm:
        rw = rw / rw.sum(-1, keepdim=True)
    assert ids.dtype == torch.int32 and wts.dtype == torch.float32
    # the chosen probabilities agree position by position (descending order, ties have equal weights)
    torch.testing.assert_close(wts, rw, atol=1e-6, rtol=1e-4)
    for t in range(min(n, 40)):
        got = set(ids[t].tolist())
        assert len(got) == k
        kth = probs[t][rid[t, -1]]
        above = set(torch.nonzero(probs[t] > kth).flatten().tolist())
        ties = torch.nonzero(probs[t] == kth).flatten().tolist()
        assert above <= got and got - above == set(ties[: k - len(above)]), f"row {t}: {sorted(got)}"
    if n > 1:
        assert ids[1].tolist() == list(range(k))


def test_router_ids_stay_in_range_for_nan_and_inf_logits():
    """A NaN or +inf logit (inf - inf) leaves no ordinary winner: the ids must still be valid, distinct experts -
    they index the expert weights in the decode kernels - and rows without such values are untouched."""
    from triton_distributed_gh_amd.ops.moe import moe_route

    E, k = 128, 8
    logits = (torch.randn(6, E, device="cuda") * 2).to(torch.bfloat16)
    clean = logits.clone()
    logits[1, 5] = float("nan")
    logits[2, :] = float("nan")
    logits[3, 7] = float("inf")
    logits[4, 9] = float("-inf")
    wts, ids = moe_route(logits, k, True)
    assert int(ids.min()) >= 0 and int(ids.max()) < E
    for t in range(6):
        assert len(set(ids[t].tolist())) == k, f"row {t}: {ids[t].tolist()}"
    w0, i0 = moe_route(clean, k, True)
    for t in (0, 5):
        assert torch.equal(ids[t], i0[t]) and torch.equal(wts[t], w0[t])
    assert torch.isfinite(wts[4]).all() and 9 not in ids[4].tolist()


def test_moe_decode_kernels_tolerate_an_expert_id_out_of_range():
    """An id beyond the expert count is clamped by the kernels (no fault); the other tokens are exact."""
    from triton_distributed_gh_amd.ops.moe import moe_decode_fwd

    n, H, I, E, topk = 3, 512, 256, 8, 2
    dev, bf = "cuda", torch.bfloat16
    x = (torch.rand(n, H, device=dev) * 2 - 1).to(bf)
    gate_up = ((torch.rand(E, 2 * I, H, device=dev) * 2 - 1) * 0.05).to(bf)
    down = ((torch.rand(E, H, I, device=dev) * 2 - 1) * 0.05).to(bf)
    ids = torch.tensor([[0, 1], [2, 3], [4, 5]], device=dev, dtype=torch.int32)
    wts = torch.full((n, topk), 0.5, device=dev)
    good = moe_decode_fwd(x, ids, wts, gate_up, down)
    bad = ids.clone()
    bad[1, 0] = 0x7fffffff
    bad[1, 1] = -5
    out = moe_decode_fwd(x, bad, wts, gate_up, down)
    torch.cuda.synchronize()
    assert torch.equal(out[0], good[0]) and torch.equal(out[2], good[2]) and torch.isfinite(out.float()).all()
