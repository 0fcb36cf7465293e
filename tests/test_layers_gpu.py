"""TP layers, PP primitive and whole models on virtual ranks of one GPU vs PyTorch fp32 goldens.

Scenarios follow the reference's layer / e2e tests: test/amd/test_tp_mlp.py, test_tp_attn.py,
test_tp_e2e.py (--check against the torch forward), test/nvidia/test_tp_moe.py, test_pp.py
(double-buffered send / recv), test_sp_decode_attn.py and test_e2e_inference.py (graph decode)."""
import os
from types import SimpleNamespace

import pytest
import torch

os.environ["RANDOM_PARAMS"] = "1"
os.environ["TDIST_NO_HF_CONFIG"] = "1"

from triton_distributed_gh_amd.layers import CommOp, SpGQAFlashDecodeAttention, TP_Attn, TP_MLP, TP_MoE  # noqa: E402
from triton_distributed_gh_amd.layers.tp_attn import _set_cos_sin_cache  # noqa: E402
from triton_distributed_gh_amd.models import AutoLLM, Engine, KV_Cache, ModelConfig  # noqa: E402
from triton_distributed_gh_amd.ops.flash_decode import decode_reference  # noqa: E402
from triton_distributed_gh_amd.utils.testing import VirtualCluster  # noqa: E402

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _lin(o, i, scale=0.05):
    return SimpleNamespace(weight=((torch.rand(o, i, device="cuda") * 2 - 1) * scale).to(BF), bias=None)


def _rel(a, b):
    return (a.float() - b.float()).abs().max().item() / max(b.float().abs().max().item(), 1e-6)


def _mlp_tree(H, I):
    return SimpleNamespace(gate_proj=_lin(I, H), up_proj=_lin(I, H), down_proj=_lin(H, I),
                           act_fn=torch.nn.functional.silu)


def _mlp_ref(x, t):
    xf = x.float()
    h = torch.nn.functional.silu(xf @ t.gate_proj.weight.float().t()) * (xf @ t.up_proj.weight.float().t())
    return h @ t.down_proj.weight.float().t()


@pytest.mark.parametrize("w,M,H,I", [(2, 512, 1024, 2048), (4, 1024, 512, 4096), (8, 2048, 1024, 2048),
                                     (4, 256, 512, 1024)])
def test_tp_mlp_modes(w, M, H, I):
    torch.manual_seed(1)
    tree = _mlp_tree(H, I)
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        mlps = [TP_MLP(rank=r, world_size=w, env=vc.envs[r]) for r in range(w)]
        for m in mlps:
            m._init_parameters(tree)

        def init(r, env):
            mlps[r]._init_ctx(max_M=M, ag_intranode_stream=[vc.side_streams[r]])
            mlps[r]._init_AR_ctx(max_M=M)
            mlps[r]._init_gemm_ar_ctx(max_M=M)

        vc.collective(init)
        x = ((torch.rand(M, H, device="cuda") * 2 - 1)).to(BF)
        ref = _mlp_ref(x, tree)
        Mr = M // w
        torch.cuda.synchronize()
        for it in range(2):
            outs = vc.run(lambda r, env: mlps[r].dist_triton_fwd(x[r * Mr:(r + 1) * Mr]))
            assert _rel(torch.cat(outs), ref) < 3e-2
            outs = vc.run(lambda r, env: mlps[r].dist_triton_AR_fwd(x))
            for o in outs:
                assert _rel(o, ref) < 3e-2
            outs = vc.run(lambda r, env: mlps[r].dist_triton_gemm_ar_fwd(x))
            for o in outs:
                assert _rel(o, ref) < 3e-2
        for m in mlps:
            m.ag_ctx.check_error()
            m.rs_ctx.check_error()
    finally:
        vc.close()


def _attn_tree(H, Hq, Hkv, D):
    norm = lambda: SimpleNamespace(weight=(1 + 0.1 * torch.randn(D, device="cuda")).to(BF), variance_epsilon=1e-6)  # noqa: E731
    return SimpleNamespace(head_dim=D, q_proj=_lin(Hq * D, H), k_proj=_lin(Hkv * D, H), v_proj=_lin(Hkv * D, H),
                           o_proj=_lin(H, Hq * D), q_norm=norm(), k_norm=norm())


def _attn_ref(x, t, pos, cos_sin, past_k=None, past_v=None):
    """fp32 reference of norm -> rope -> causal GQA attention -> o_proj.  x [B, S, H]."""
    B, S, H = x.shape
    D = t.head_dim
    xf = x.float()
    q = (xf @ t.q_proj.weight.float().t()).view(B, S, -1, D)
    k = (xf @ t.k_proj.weight.float().t()).view(B, S, -1, D)
    v = (xf @ t.v_proj.weight.float().t()).view(B, S, -1, D)

    def rms(z, n):
        return z * torch.rsqrt(z.pow(2).mean(-1, keepdim=True) + 1e-6) * n.weight.float()

    def rope(z):
        c = cos_sin[0][pos].unsqueeze(2)
        s = cos_sin[1][pos].unsqueeze(2)
        z1, z2 = z[..., : D // 2], z[..., D // 2:]
        return torch.cat([z1 * c - z2 * s, z2 * c + z1 * s], -1)

    q, k = rope(rms(q, t.q_norm)), rope(rms(k, t.k_norm))
    if past_k is not None:
        k, v = torch.cat([past_k, k], 1), torch.cat([past_v, v], 1)
    Hq, Hkv = q.shape[2], k.shape[2]
    kk = k.repeat_interleave(Hq // Hkv, dim=2)
    vv = v.repeat_interleave(Hq // Hkv, dim=2)
    s = torch.einsum("bqhd,bkhd->bhqk", q, kk) / D ** 0.5
    Lk = k.shape[1]
    mask = torch.arange(Lk, device=x.device)[None, :] <= (torch.arange(S, device=x.device)[:, None] + (Lk - S))
    p = torch.softmax(s.masked_fill(~mask, float("-inf")), -1)
    o = torch.einsum("bhqk,bkhd->bqhd", p, vv).reshape(B, S, -1)
    return o @ t.o_proj.weight.float().t(), k, v


@pytest.mark.parametrize("w,B,S,H,Hq,Hkv,D", [(2, 4, 64, 1024, 8, 4, 128), (4, 4, 32, 512, 16, 8, 64),
                                              (4, 8, 32, 512, 8, 8, 64)])
def test_tp_attn_prefill_and_decode(w, B, S, H, Hq, Hkv, D):
    torch.manual_seed(2)
    tree = _attn_tree(H, Hq, Hkv, D)
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device="cuda").float() / D))
    cos_sin = _set_cos_sin_cache(inv, 256)
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        attns = [TP_Attn(rank=r, world_size=w, env=vc.envs[r]) for r in range(w)]
        for a in attns:
            a._init_parameters(tree)

        def init(r, env):
            attns[r]._init_ctx(max_M=B * S, ag_intranode_stream=[vc.side_streams[r]])
            attns[r]._init_AR_ctx(max_M=B * S)
            attns[r]._init_gemm_ar_ctx(max_M=B * S)

        vc.collective(init)
        for mode in ("dist_triton_fwd", "dist_triton_AR_fwd", "dist_triton_gemm_ar_fwd"):
            kvs = [KV_Cache(1, B, 256, Hkv, D, BF, w) for _ in range(w)]
            x = (torch.rand(B, S, H, device="cuda") * 2 - 1).to(BF)
            pos = torch.arange(S, device="cuda")[None].repeat(B, 1)
            ref, pk, pv = _attn_ref(x, tree, pos, cos_sin)
            sharded = mode == "dist_triton_fwd"
            per = B // w
            torch.cuda.synchronize()
            outs = vc.run(lambda r, env: getattr(attns[r], mode)(x[r * per:(r + 1) * per] if sharded else x, pos, cos_sin,
                                                                 kvs[r], 0))
            got = torch.cat(outs) if sharded else outs[0]
            assert _rel(got, ref) < 3e-2, mode
            for kv in kvs:
                kv.kv_offset.fill_(S)
            # one decode step on top of the prefilled cache
            x1 = (torch.rand(B, 1, H, device="cuda") * 2 - 1).to(BF)
            pos1 = torch.full((B, 1), S, device="cuda", dtype=torch.long)
            ref1, _, _ = _attn_ref(x1, tree, pos1, cos_sin, pk, pv)
            torch.cuda.synchronize()
            outs = vc.run(lambda r, env: getattr(attns[r], mode)(x1[r * per:(r + 1) * per] if sharded else x1, pos1,
                                                                 cos_sin, kvs[r], 0))
            got = torch.cat(outs) if sharded else outs[0]
            assert _rel(got, ref1) < 3e-2, mode + " decode"
    finally:
        vc.close()


@pytest.mark.parametrize("w,T,H,I,E,topk", [(2, 256, 512, 512, 8, 2), (4, 512, 1024, 1024, 16, 4)])
def test_tp_moe_matches_expert_loop(w, T, H, I, E, topk):
    torch.manual_seed(3)
    tree = SimpleNamespace(num_experts=E, top_k=topk, norm_topk_prob=True, gate=_lin(E, H, 1.0),
                           experts=[_mlp_tree(H, I) for _ in range(E)])
    x = (torch.rand(T, H, device="cuda") * 2 - 1).to(BF)
    logits = x @ tree.gate.weight.t()
    wts, ids = torch.topk(torch.softmax(logits, 1, dtype=torch.float), topk, dim=-1)
    wts = wts / wts.sum(-1, keepdim=True)
    ref = torch.zeros(T, H, device="cuda")
    for e in range(E):
        tok, k = torch.where(ids == e)
        if tok.numel():
            ref.index_add_(0, tok, _mlp_ref(x[tok], tree.experts[e]) * wts[tok, k, None])
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        moes = [TP_MoE(rank=r, world_size=w, env=vc.envs[r]) for r in range(w)]
        for m in moes:
            m._init_parameters(tree)
        vc.collective(lambda r, env: moes[r]._init_ctx(T, ag_streams=[vc.side_streams[r]]))
        for m in moes:
            m.dec_per = 0       # this test is about the AG + grouped GEMM / grouped GEMM + RS pair: no gather / all-reduce form
        per = T // w
        torch.cuda.synchronize()
        for it in range(2):
            outs = vc.run(lambda r, env: moes[r].dist_triton_fwd(x[r * per:(r + 1) * per]))
            assert _rel(torch.cat(outs), ref) < 4e-2
    finally:
        vc.close()


@pytest.mark.parametrize("w,T,H,I,E,topk", [(2, 8, 512, 512, 8, 2), (4, 16, 1024, 1024, 16, 4), (2, 2, 512, 512, 8, 8),
                                            (2, 64, 512, 512, 8, 2), (4, 96, 1024, 1024, 16, 4), (2, 200, 512, 512, 8, 2)])
def test_tp_moe_decode_path_on_several_ranks(w, T, H, I, E, topk):
    """A decode-sized batch on a tensor-parallel MoE layer: rows all-gathered, every rank runs all tokens against its
    slice of the experts with the decode launches (per pair, or - the cases with 64 and 96 tokens - grouped by expert
    in blocks of sixteen rows), partial sums all-reduced - eagerly and replayed from a graph, with a larger
    (grouped-GEMM) call in between."""
    from triton_distributed_gh_amd.ops import moe as moe_ops

    torch.manual_seed(5)
    tree = SimpleNamespace(num_experts=E, top_k=topk, norm_topk_prob=True, gate=_lin(E, H, 1.0),
                           experts=[_mlp_tree(H, I) for _ in range(E)])

    def reference(x):
        logits = x @ tree.gate.weight.t()
        wts, ids = torch.topk(torch.softmax(logits, 1, dtype=torch.float), topk, dim=-1)
        wts = wts / wts.sum(-1, keepdim=True)
        ref = torch.zeros(x.shape[0], H, device="cuda")
        for e in range(E):
            tok, k = torch.where(ids == e)
            if tok.numel():
                ref.index_add_(0, tok, _mlp_ref(x[tok], tree.experts[e]) * wts[tok, k, None])
        return ref

    calls = {"n": 0}
    real, real_mid = moe_ops.moe_decode_fwd, moe_ops.moe_group_mid_fwd

    def counted(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    def counted_mid(*a, **k):          # (the case with 200 tokens: 64-row blocks on the mid-M GEMM body)
        calls["n"] += 1
        return real_mid(*a, **k)

    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        import triton_distributed_gh_amd.layers.tp_moe as tp_moe_mod
        tp_moe_mod.moe_decode_fwd = counted
        tp_moe_mod.moe_group_mid_fwd = counted_mid
        moes = [TP_MoE(rank=r, world_size=w, env=vc.envs[r]) for r in range(w)]
        for m in moes:
            m._init_parameters(tree)
        vc.collective(lambda r, env: moes[r]._init_ctx(256, ag_streams=[vc.side_streams[r]]))
        per = T // w
        xs = [torch.zeros(per, H, device="cuda", dtype=BF) for _ in range(w)]

        def fill():
            x = (torch.rand(T, H, device="cuda") * 2 - 1).to(BF)
            for r in range(w):
                xs[r].copy_(x[r * per:(r + 1) * per])
            torch.cuda.synchronize()
            return x

        for it in range(3):
            x = fill()
            outs = vc.run(lambda r, env: moes[r].dist_triton_fwd(xs[r]))
            assert _rel(torch.cat(outs), reference(x)) < 4e-2, f"eager {it}"
        assert calls["n"] == 3 * w, "the decode launches ran on every rank"
        # a prefill-sized call (grouped GEMM pair) between decode calls
        big = (torch.rand(256, H, device="cuda") * 2 - 1).to(BF)
        torch.cuda.synchronize()
        saved = [m.dec_per for m in moes]
        for m in moes:
            m.dec_per = 0                  # (no row budget for the gather / all-reduce form: the grouped-GEMM pair runs)
        outs = vc.run(lambda r, env: moes[r].dist_triton_fwd(big[r * (256 // w):(r + 1) * (256 // w)].contiguous()))
        for m, v in zip(moes, saved):
            m.dec_per = v
        assert _rel(torch.cat(outs), reference(big)) < 4e-2
        assert calls["n"] == 3 * w, "the prefill-sized call took the grouped-GEMM pair"
        graphs, static = [], [None] * w
        for r in range(w):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=vc.streams[r]):
                static[r] = moes[r].dist_triton_fwd(xs[r])
            graphs.append(g)
        for it in range(4):
            x = fill()
            vc.run(lambda r, env: graphs[r].replay())
            for m in moes:
                m.dec_ar_ctx.check_error()
            assert _rel(torch.cat(static), reference(x)) < 4e-2, f"replay {it}"
    finally:
        tp_moe_mod.moe_decode_fwd = real
        tp_moe_mod.moe_group_mid_fwd = real_mid
        vc.close()


@pytest.mark.parametrize("w,T,H,I,E,topk", [(2, 256, 512, 512, 8, 2), (4, 64, 512, 512, 8, 2)])
def test_tp_moe_replays_from_graph(w, T, H, I, E, topk):
    """The whole tensor-parallel MoE layer - router, low-latency all-gather of the routing data, AG + grouped
    GEMM, activation, grouped GEMM + top-k reduce + reduce-scatter - captured once per rank and replayed on
    fresh inputs: all three collectives number their calls on the device."""
    torch.manual_seed(11)
    tree = SimpleNamespace(num_experts=E, top_k=topk, norm_topk_prob=True, gate=_lin(E, H, 1.0),
                           experts=[_mlp_tree(H, I) for _ in range(E)])

    def reference(x):
        logits = x @ tree.gate.weight.t()
        wts, ids = torch.topk(torch.softmax(logits, 1, dtype=torch.float), topk, dim=-1)
        wts = wts / wts.sum(-1, keepdim=True)
        ref = torch.zeros(T, H, device="cuda")
        for e in range(E):
            tok, k = torch.where(ids == e)
            if tok.numel():
                ref.index_add_(0, tok, _mlp_ref(x[tok], tree.experts[e]) * wts[tok, k, None])
        return ref

    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        moes = [TP_MoE(rank=r, world_size=w, env=vc.envs[r]) for r in range(w)]
        for m in moes:
            m._init_parameters(tree)
        vc.collective(lambda r, env: moes[r]._init_ctx(T, ag_streams=[vc.side_streams[r]]))
        for m in moes:
            m.dec_per = 0       # this test is about the AG + grouped GEMM / grouped GEMM + RS pair: no gather / all-reduce form
        per = T // w
        xs = [torch.zeros(per, H, device="cuda", dtype=BF) for _ in range(w)]

        def fill():
            x = (torch.rand(T, H, device="cuda") * 2 - 1).to(BF)
            for r in range(w):
                xs[r].copy_(x[r * per:(r + 1) * per])
            torch.cuda.synchronize()
            return x

        x = fill()
        outs = vc.run(lambda r, env: moes[r].dist_triton_fwd(xs[r]))       # eager: an odd number of calls before capture
        assert _rel(torch.cat(outs), reference(x)) < 4e-2
        graphs, static = [], [None] * w
        for r in range(w):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=vc.streams[r]):
                static[r] = moes[r].dist_triton_fwd(xs[r])
            graphs.append(g)
        for it in range(4):
            x = fill()
            vc.run(lambda r, env: graphs[r].replay())
            for m in moes:
                m.ag_ctx.check_error()
                m.rs_ctx.check_error()
            assert _rel(torch.cat(static), reference(x)) < 4e-2, f"replay {it}"
            if it == 1:                                                      # an eager call between replays
                outs = vc.run(lambda r, env: moes[r].dist_triton_fwd(xs[r]))
                assert _rel(torch.cat(outs), reference(x)) < 4e-2
    finally:
        vc.close()


def test_pp_commop_double_buffered_send_recv():
    """Rank 0 sends 6 activations to rank 1 through 2 alternating buffers (pull + signal handshake)."""
    w, T, Hd = 2, 128, 512
    vc = VirtualCluster(w, heap_bytes=1 << 26)
    try:
        ops = vc.collective(lambda r, env: CommOp(T, Hd, r, w, None, BF, num_buffers=2, env=env))
        msgs = [torch.randn(T, Hd, device="cuda").to(BF) for _ in range(6)]
        recv = [torch.empty(T, Hd, device="cuda", dtype=BF) for _ in range(6)]
        torch.cuda.synchronize()

        def fn(r, env):
            for i, m in enumerate(msgs):
                buf = i % 2
                if r == 0:
                    ops[0].wait_signal(0, buf, 0)             # slot free
                    ops[0].get_buffer(buf).copy_(m)
                    ops[0].set_signal(0, buf, 1)              # data ready
                else:
                    ops[1].wait_signal(0, buf, 1)
                    ops[1].read(0, buf, recv[i], sm=4)
                    ops[1].set_signal(0, buf, 0)              # hand the slot back

        vc.run(fn)
        for m, g in zip(msgs, recv):
            assert torch.equal(m, g)
    finally:
        vc.close()


@pytest.mark.parametrize("w,batch,Hq,Hkv,D,kv_len", [(2, 2, 16, 2, 128, 700), (4, 1, 32, 4, 128, 1500)])
def test_sp_flash_decode_layer(w, batch, Hq, Hkv, D, kv_len):
    torch.manual_seed(4)
    page = 16
    per_rank = (kv_len + w - 1) // w
    pages = (per_rank + page - 1) // page
    q = torch.randn(batch, Hq, D, device="cuda").to(BF)
    k_all = torch.randn(batch, kv_len, Hkv, D, device="cuda").to(BF)
    v_all = torch.randn(batch, kv_len, Hkv, D, device="cuda").to(BF)
    scale = D ** -0.5
    # reference over the whole sequence
    kc = k_all.reshape(batch * kv_len, 1, Hkv, D)
    vcache = v_all.reshape(batch * kv_len, 1, Hkv, D)
    table = torch.arange(batch * kv_len, device="cuda", dtype=torch.int32).view(batch, kv_len)
    ref = decode_reference(q, kc, vcache, torch.full((batch,), kv_len, device="cuda", dtype=torch.int32), table, scale)
    vc = VirtualCluster(w, heap_bytes=1 << 26)
    try:
        layers = vc.collective(lambda r, env: SpGQAFlashDecodeAttention(r, 0, w, 1, Hq, Hkv, D, D, page_size=page,
                                                                        scale=scale, max_allowed_batch=batch, env=env))
        lens = torch.zeros(w, batch, dtype=torch.int32, device="cuda")
        shards = []
        for r in range(w):
            lo, hi = r * per_rank, min(kv_len, (r + 1) * per_rank)
            n = max(hi - lo, 0)
            lens[r] = n
            kr = torch.zeros(batch * pages, page, Hkv, D, device="cuda", dtype=BF)
            vr = torch.zeros_like(kr)
            for b in range(batch):
                kr[b * pages:(b + 1) * pages].view(-1, Hkv, D)[:n] = k_all[b, lo:hi]
                vr[b * pages:(b + 1) * pages].view(-1, Hkv, D)[:n] = v_all[b, lo:hi]
            bt = torch.arange(batch * pages, device="cuda", dtype=torch.int32).view(batch, pages)
            shards.append((kr, vr, bt))
        torch.cuda.synchronize()
        for it in range(3):
            outs = vc.run(lambda r, env: layers[r](q, shards[r][0], shards[r][1], lens, shards[r][2]))
            for o in outs:
                assert _rel(o, ref) < 2e-2
    finally:
        vc.close()


@pytest.mark.parametrize("name,mode,w", [("tdist/tiny-dense", "triton_dist", 2), ("tdist/tiny-dense", "triton_dist", 4),
                                         ("tdist/tiny-dense", "triton_dist_AR", 2),
                                         ("tdist/tiny-dense", "triton_dist_gemm_ar", 4),
                                         ("tdist/tiny-moe", "triton_dist", 2)])
def test_model_modes_match_single_rank(name, mode, w):
    bsz, prompt, L = 8, 32, 256
    torch.manual_seed(5)
    ids = torch.randint(0, 1000, (bsz, prompt), device="cuda")
    pos = torch.arange(prompt, device="cuda")[None].repeat(bsz, 1)
    g = AutoLLM.from_pretrained(ModelConfig(model_name=name, max_length=L, dtype=BF, rank=0, world_size=1))
    gkv = KV_Cache(g.num_layers, bsz, L, g.num_key_value_heads, g.head_dim, BF, 1)
    gold = [g.inference(ids, pos, gkv)]
    gkv.kv_offset.fill_(prompt)
    tok = gold[0][:, -1].argmax(-1, keepdim=True)
    gold.append(g.inference(tok, gkv.get_kv_len()[:, None].long(), gkv))
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        models = [AutoLLM.from_pretrained(ModelConfig(model_name=name, max_length=L, dtype=BF, rank=r, world_size=w),
                                          env=vc.envs[r]) for r in range(w)]
        kvs = [KV_Cache(g.num_layers, bsz, L, g.num_key_value_heads, g.head_dim, BF, w) for _ in range(w)]
        for m in models:
            m.set_fwd(mode)
        init = {"triton_dist": lambda m: m.init_triton_dist_ctx(max_M=bsz * prompt),
                "triton_dist_AR": lambda m: m.init_triton_dist_AR_ctx(max_M=bsz * prompt),
                "triton_dist_gemm_ar": lambda m: m.init_triton_dist_gemm_ar_ctx(max_M=bsz * prompt)}[mode]
        vc.collective(lambda r, env: init(models[r]))
        sharded = mode == "triton_dist"
        per = bsz // w
        sh = (lambda x, r: x[r * per:(r + 1) * per]) if sharded else (lambda x, r: x)
        torch.cuda.synchronize()
        outs = vc.run(lambda r, env: models[r].inference(sh(ids, r), pos, kvs[r]))
        got = torch.cat(outs) if sharded else outs[0]
        assert _rel(got, gold[0]) < 5e-2
        for kv in kvs:
            kv.kv_offset.fill_(prompt)
        outs = vc.run(lambda r, env: models[r].inference(sh(tok, r), kvs[r].get_kv_len()[:, None].long(), kvs[r]))
        got = torch.cat(outs) if sharded else outs[0]
        assert _rel(got, gold[1]) < 5e-2
    finally:
        vc.close()


@pytest.mark.parametrize("mode", ["triton_dist_AR", "triton_dist_gemm_ar", "triton_dist", "triton_dist-256tile",
                                  "triton_dist-moe"])
def test_fused_backend_decode_replays_from_graph(mode, monkeypatch):
    """World size 2: each rank captures ONE decode step and replays it; the all-reduce kernels and the
    low-latency forms of GEMM + all-reduce / AG + GEMM / GEMM + RS advance their call number on the device,
    so every replay matches the single-rank model.  ``triton_dist-256tile`` switches the low-latency forms
    off, so the two-row shards reach the 256x256-tile AG + GEMM / GEMM + RS kernels as tiles that straddle
    ranks (the configuration whose replay faulted while a memset node preceded every launch)."""
    if mode.endswith("-256tile"):
        from triton_distributed_gh_amd.ops import tp_ll

        monkeypatch.setattr(tp_ll, "LL_MAX_M", 0)
        mode = "triton_dist"
    name, w, bsz, prompt, L, steps = "tdist/tiny-dense", 2, 4, 16, 128, 4
    tol = 5e-2
    if mode.endswith("-moe"):           # the MoE model: its layers capture the MoE pair and the routing all-gather
        name, mode = "tdist/tiny-moe", "triton_dist"
        tol = 8e-2                      # (bf16 partials of eight experts per token, four steps of cache drift)
    sharded = mode == "triton_dist"
    per = bsz // w
    sh = (lambda x, r: x[r * per:(r + 1) * per].contiguous()) if sharded else (lambda x, r: x)
    torch.manual_seed(7)
    ids = torch.randint(0, 1000, (bsz, prompt), device="cuda")
    pos = torch.arange(prompt, device="cuda")[None].repeat(bsz, 1)
    g = AutoLLM.from_pretrained(ModelConfig(model_name=name, max_length=L, dtype=BF, rank=0, world_size=1))
    gkv = KV_Cache(g.num_layers, bsz, L, g.num_key_value_heads, g.head_dim, BF, 1)
    tok = g.inference(ids, pos, gkv)[:, -1].argmax(-1, keepdim=True)
    gkv.kv_offset.fill_(prompt)
    toks, gold = [tok], []
    for _ in range(steps):
        gold.append(g.inference(toks[-1], gkv.get_kv_len()[:, None].long(), gkv))
        gkv.inc_offset()
        toks.append(gold[-1][:, -1].argmax(-1, keepdim=True))
    vc = VirtualCluster(w, heap_bytes=1 << 28)
    try:
        models = [AutoLLM.from_pretrained(ModelConfig(model_name=name, max_length=L, dtype=BF, rank=r, world_size=w),
                                          env=vc.envs[r]) for r in range(w)]
        kvs = [KV_Cache(g.num_layers, bsz, L, g.num_key_value_heads, g.head_dim, BF, w) for _ in range(w)]
        for m in models:
            m.set_fwd(mode)
        if sharded:
            vc.collective(lambda r, env: models[r].init_triton_dist_ctx(max_M=bsz * prompt))
        elif mode == "triton_dist_gemm_ar":
            vc.collective(lambda r, env: models[r].init_triton_dist_gemm_ar_ctx(max_M=bsz * prompt))
        else:
            vc.collective(lambda r, env: models[r].init_triton_dist_AR_ctx(max_M=bsz * prompt))
        torch.cuda.synchronize()
        vc.run(lambda r, env: models[r].inference(sh(ids, r), pos, kvs[r]))            # eager prefill
        for kv in kvs:
            kv.kv_offset.fill_(prompt)
        s_tok = [torch.zeros_like(sh(tok, r)) for r in range(w)]
        s_pos = [torch.zeros((bsz, 1), dtype=torch.long, device="cuda") for _ in range(w)]
        graphs, logits = [], [None] * w
        # warm-up step outside capture (lazy tables), undone by not advancing the KV offset
        for r in range(w):
            s_tok[r].copy_(sh(toks[0], r))
            s_pos[r].copy_(kvs[r].get_kv_len()[:, None].long())
        torch.cuda.synchronize()
        vc.run(lambda r, env: models[r].inference(s_tok[r], s_pos[r], kvs[r]))
        for r in range(w):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=vc.streams[r]):
                logits[r] = models[r].inference(s_tok[r], s_pos[r], kvs[r])
            graphs.append(gr)
        for step in range(steps):
            for r in range(w):
                s_tok[r].copy_(sh(toks[step], r))
                s_pos[r].copy_(kvs[r].get_kv_len()[:, None].long())
            torch.cuda.synchronize()
            vc.run(lambda r, env: graphs[r].replay())
            got = torch.cat(logits) if sharded else logits[0]
            assert _rel(got, gold[step]) < tol
            for r in range(w):
                if not sharded:
                    assert _rel(logits[r], gold[step]) < tol
                kvs[r].inc_offset()
    finally:
        vc.close()


def test_engine_graph_decode_matches_eager():
    cfg = ModelConfig(model_name="tdist/tiny-dense", max_length=128, dtype=BF, rank=0, world_size=1)
    eng = Engine(cfg)
    prompt = torch.randint(0, 1000, (4, 16))
    eng.no_graph = True
    eager = eng.serve(prompt, gen_len=8)
    eng.no_graph = False
    graphed = eng.serve(prompt, gen_len=8)
    assert torch.equal(eager, graphed)


def test_decode_blocks_fold_their_norm(monkeypatch):
    """A decode step in the modes whose first projections are local GEMMs calls the fused norm + GEMM once per
    attention block and once per MLP block (no separate norm launch), prefill keeps the separate kernels, and the
    logits match the torch mode."""
    from triton_distributed_gh_amd.layers import tp_attn, tp_mlp

    calls = {"attn": 0, "mlp": 0}
    real = tp_attn.gemm_nt_norm

    def spy(kind):
        def f(*a, **k):
            calls[kind] += 1
            return real(*a, **k)
        return f

    monkeypatch.setattr(tp_attn, "gemm_nt_norm", spy("attn"))
    monkeypatch.setattr(tp_mlp, "gemm_nt_norm", spy("mlp"))
    bsz, prompt, L = 2, 16, 128
    torch.manual_seed(11)
    m = AutoLLM.from_pretrained(ModelConfig(model_name="tdist/tiny-dense", max_length=L, dtype=BF, rank=0, world_size=1))
    ids = torch.randint(0, 1000, (bsz, prompt), device="cuda")
    pos = torch.arange(prompt, device="cuda")[None].repeat(bsz, 1)
    outs = {}
    for mode in ("torch", "triton_dist_AR"):
        m.set_fwd(mode)
        kv = KV_Cache(m.num_layers, bsz, L, m.num_key_value_heads, m.head_dim, BF, 1)
        first = m.inference(ids, pos, kv)
        assert calls == {"attn": 0, "mlp": 0}, "prefill-sized inputs keep the norm kernel"
        kv.kv_offset.fill_(prompt)
        tok = first[:, -1].argmax(-1, keepdim=True)
        outs[mode] = m.inference(tok, kv.get_kv_len()[:, None].long(), kv)
    assert calls == {"attn": m.num_layers, "mlp": m.num_layers}, calls
    assert _rel(outs["triton_dist_AR"], outs["torch"]) < 5e-2
