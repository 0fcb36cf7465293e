"""Flash-decode (split-KV / fused / intra-rank + inter-rank merge), varlen flash attention and
sequence-parallel all-gather attention vs fp32 PyTorch references (scenarios of the reference's
test_decode_attn.py, test_sp_decode_attn.py and test_sp_ag_attention_intra_node.py)."""
import math

import pytest
import torch

from triton_distributed_gh_amd.ops import flash_decode as fd
from triton_distributed_gh_amd.ops.flash_attn import (attention_reference, flash_attn_varlen_fwd,
                                                      flash_attn_with_kvcache)
from triton_distributed_gh_amd.ops.sp_ag_attention import (create_sp_ag_attention_context_intra_node,
                                                           fused_sp_ag_attn_intra_node, shard_positions)
from triton_distributed_gh_amd.utils.testing import VirtualCluster

pytestmark = pytest.mark.gpu


def _paged_cache(batch, kv_lens, page, kv_heads, D, dtype, extra_pages=3):
    max_pages = max((int(l) + page - 1) // page for l in kv_lens) + 1
    total_pages = batch * max_pages + extra_pages
    k = (torch.randn(total_pages, page, kv_heads, D, device="cuda") * 0.5).to(dtype)
    v = (torch.randn(total_pages, page, kv_heads, D, device="cuda") * 0.5).to(dtype)
    perm = torch.randperm(total_pages, device="cuda")[: batch * max_pages].int().view(batch, max_pages)
    return k, v, perm.contiguous()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("batch,q_heads,kv_heads,D,page,kv_split,soft_cap", [
    (2, 32, 4, 128, 16, 4, 0.0),
    (3, 24, 3, 128, 1, -1, 0.0),
    (1, 8, 8, 64, 64, 3, 0.0),
    (2, 40, 2, 128, 32, 8, 30.0),
    (1, 16, 2, 256, 16, 2, 0.0),
])
def test_flash_decode_variants(dtype, batch, q_heads, kv_heads, D, page, kv_split, soft_cap):
    kv_lens = torch.tensor([1000, 37, 513][:batch], dtype=torch.int32, device="cuda")
    k, v, table = _paged_cache(batch, kv_lens.tolist(), page, kv_heads, D, dtype)
    q = torch.randn(batch, q_heads, D, device="cuda").to(dtype)
    scale = 1.0 / math.sqrt(D)
    ref, ref_lse = fd.decode_reference(q, k, v, kv_lens, table, scale, soft_cap, return_lse=True)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    o1 = fd.gqa_fwd_batch_decode(q, k, v, None, None, kv_lens, table, scale, soft_cap, kv_split=kv_split)
    o2 = fd.gqa_fwd_batch_decode_persistent(q, k, v, None, None, kv_lens, table, scale, soft_cap, kv_split=kv_split)
    o3 = fd.gqa_fwd_batch_decode_intra_rank(q, k, v, None, None, kv_lens, table, scale, soft_cap, kv_split=kv_split)
    torch.cuda.synchronize()
    assert (o1.float() - ref).abs().max().item() < tol
    assert (o2.float() - ref).abs().max().item() < tol
    assert (o3[..., :D] - ref).abs().max().item() < tol
    assert (o3[..., D] - ref_lse).abs().max().item() < 2e-2
    # a second fused launch must find its tickets reset
    o4 = fd.gqa_fwd_batch_decode_persistent(q, k, v, None, None, kv_lens, table, scale, soft_cap, kv_split=kv_split)
    assert torch.equal(o2, o4)


def test_flash_decode_zero_length_and_inter_rank_merge():
    R, batch, q_heads, kv_heads, D, page = 4, 2, 16, 2, 128, 16
    dtype = torch.bfloat16
    lens = torch.tensor([[300, 0], [17, 90], [0, 0], [129, 64]], dtype=torch.int32, device="cuda")
    q = torch.randn(batch, q_heads, D, device="cuda").to(dtype)
    scale = 1.0 / math.sqrt(D)
    parts, ks, vs = [], [], []
    for r in range(R):
        k, v, table = _paged_cache(batch, [max(int(x), 1) for x in lens[r]], page, kv_heads, D, dtype)
        parts.append(fd.gqa_fwd_batch_decode_intra_rank(q, k, v, None, None, lens[r].contiguous(), table, scale))
        for b in range(batch):
            L = int(lens[r, b])
            npg = (L + page - 1) // page
            ks.append((b, k[table[b, :npg].long()].reshape(-1, kv_heads, D)[:L]))
            vs.append((b, v[table[b, :npg].long()].reshape(-1, kv_heads, D)[:L]))
    out = fd.gqa_fwd_batch_decode_inter_rank_combine(torch.stack(parts), lens, dtype=dtype)
    for b in range(batch):
        kk = torch.cat([t for bb, t in ks if bb == b]).float()
        vv = torch.cat([t for bb, t in vs if bb == b]).float()
        s = torch.einsum("hgd,lhd->hgl", q[b].float().view(kv_heads, -1, D), kk) * scale
        ref = torch.einsum("hgl,lhd->hgd", torch.softmax(s, -1), vv).reshape(q_heads, D)
        assert (out[b].float() - ref).abs().max().item() < 2e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("seqs_q,seqs_k,Hq,Hkv,D,causal", [
    ([128], [128], 4, 4, 128, True),
    ([200, 77, 333], [200, 77, 333], 8, 2, 128, True),
    ([64, 130], [300, 130], 6, 3, 64, True),
    ([100, 260], [150, 260], 4, 1, 128, False),
])
def test_flash_attn_varlen(dtype, seqs_q, seqs_k, Hq, Hkv, D, causal):
    cu_q = torch.tensor([0] + list(torch.tensor(seqs_q).cumsum(0)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(torch.tensor(seqs_k).cumsum(0)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(seqs_q), Hq, D, device="cuda").to(dtype)
    k = torch.randn(sum(seqs_k), Hkv, D, device="cuda").to(dtype)
    v = torch.randn(sum(seqs_k), Hkv, D, device="cuda").to(dtype)
    out, lse = flash_attn_varlen_fwd(q, k, v, cu_q, cu_k, max(seqs_q), causal=causal, return_lse=True)
    ref = attention_reference(q, k, v, cu_q.tolist(), cu_k.tolist(), causal)
    torch.cuda.synchronize()
    tol = 3e-2 if dtype == torch.bfloat16 else 5e-3
    assert (out.float() - ref).abs().max().item() < tol
    assert torch.isfinite(lse).all()


def test_flash_attn_with_kvcache_prefill_then_decode():
    B, Hq, Hkv, D, max_len = 2, 8, 2, 128, 512
    dtype = torch.bfloat16
    kc = torch.zeros(B, max_len, Hkv, D, device="cuda", dtype=dtype)
    vc = torch.zeros_like(kc)
    S = 96
    q = torch.randn(B, S, Hq, D, device="cuda").to(dtype)
    k = torch.randn(B, S, Hkv, D, device="cuda").to(dtype)
    v = torch.randn(B, S, Hkv, D, device="cuda").to(dtype)
    lens = torch.zeros(B, dtype=torch.int32, device="cuda")
    o = flash_attn_with_kvcache(q, kc, vc, k, v, cache_seqlens=lens)
    for b in range(B):
        ref = attention_reference(q[b], k[b], v[b], [0, S], [0, S], True)
        assert (o[b].float() - ref).abs().max().item() < 3e-2
    assert torch.equal(kc[:, :S], k) and torch.equal(vc[:, :S], v)
    q1 = torch.randn(B, 1, Hq, D, device="cuda").to(dtype)
    k1 = torch.randn(B, 1, Hkv, D, device="cuda").to(dtype)
    v1 = torch.randn(B, 1, Hkv, D, device="cuda").to(dtype)
    lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
    o1 = flash_attn_with_kvcache(q1, kc, vc, k1, v1, cache_seqlens=lens)
    for b in range(B):
        kk = torch.cat([k[b], k1[b]])
        vv = torch.cat([v[b], v1[b]])
        ref = attention_reference(q1[b], kk, vv, [0, 1], [0, S + 1], True)
        assert (o1[b].float() - ref).abs().max().item() < 3e-2


@pytest.mark.parametrize("w,zigzag,causal", [(2, True, True), (4, True, True), (4, False, True), (8, True, True),
                                               (2, False, False)])
def test_sp_ag_attention(w, zigzag, causal):
    Hq, Hkv, D = 4, 2, 128
    dtype = torch.bfloat16
    full = [64 * 2 * w, 128 * 2 * w]                 # per-sequence full lengths (multiples of 64 per half-block)
    shard = [L // w for L in full]
    cu_k = torch.tensor([0, full[0], full[0] + full[1]], dtype=torch.int32, device="cuda")
    cu_q = torch.tensor([0, shard[0], shard[0] + shard[1]], dtype=torch.int32, device="cuda")
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ctxs = vc.collective(lambda r, env: create_sp_ag_attention_context_intra_node(
            2, Hq, Hkv, max(full), max(shard), D, dtype, dtype, r, w, ag_stream=vc.side_streams[r], env=env))
        for it in range(3):
            qf = torch.randn(sum(full), Hq, D, device="cuda").to(dtype)
            kf = torch.randn(sum(full), Hkv, D, device="cuda").to(dtype)
            vf = torch.randn(sum(full), Hkv, D, device="cuda").to(dtype)
            ref = attention_reference(qf, kf, vf, cu_k.tolist(), cu_k.tolist(), causal)

            def rows(r):
                idx = []
                for b in range(2):
                    pos = shard_positions([0, shard[b]], r, w, zigzag)
                    idx.append(pos + int(cu_k[b]))
                return torch.cat(idx).cuda()

            qs = [qf[rows(r)].contiguous() for r in range(w)]
            ks = [kf[rows(r)].contiguous() for r in range(w)]
            vs = [vf[rows(r)].contiguous() for r in range(w)]
            outs = [torch.empty_like(qs[r]) for r in range(w)]
            torch.cuda.synchronize()
            vc.run(lambda r, env: fused_sp_ag_attn_intra_node(ctxs[r], qs[r], ks[r], vs[r], outs[r], cu_q, cu_k,
                                                              max(shard), max(full), r, w, causal, zigzag))
            for r in range(w):
                ctxs[r].check_error()
                err = (outs[r].float() - ref[rows(r)]).abs().max().item()
                assert err < 3e-2, f"rank {r} iter {it}: {err}"
    finally:
        vc.close()


@pytest.mark.parametrize("B,S,Hq,Hkv,D,norm", [(1, 1, 32, 8, 128, True), (3, 1, 8, 8, 64, False), (2, 5, 16, 4, 128, True),
                                                (1, 7, 4, 2, 256, True)])
def test_qkv_prep_matches_separate_kernels(B, S, Hq, Hkv, D, norm):
    """Fused per-head norm + rope + cache append + attended length vs rms_norm / apply_rope_ / kv_append."""
    from triton_distributed_gh_amd.ops import elementwise

    torch.manual_seed(B * 100 + S * 10 + Hq)
    dev, bf = "cuda", torch.bfloat16
    max_len, W = 64, (Hq + 2 * Hkv) * D
    big = (torch.rand(B, S, W + 64, device=dev) * 2 - 1).to(bf)
    qkv = big[:, :, :W]                                            # strided rows
    qw = (torch.rand(D, device=dev) + 0.5).to(bf) if norm else None
    kw = (torch.rand(D, device=dev) + 0.5).to(bf) if norm else None
    cos, sin = elementwise.rope_cache(256, D, device=dev)
    offs = torch.randint(0, max_len - S, (B,), device=dev, dtype=torch.int32)
    pos = offs.long()[:, None] + torch.arange(S, device=dev)[None]
    kc = torch.zeros(B, max_len, Hkv, D, device=dev, dtype=bf)
    vc = torch.zeros_like(kc)
    total = torch.zeros(B, dtype=torch.int32, device=dev)
    assert elementwise.qkv_prep_supported(qkv, kc, vc, D)
    q = elementwise.qkv_prep(qkv, qw, kw, 1e-6, 1e-6, pos, cos, sin, kc, vc, offs, total, Hq, Hkv, D)
    # reference: the separate ops
    qr, kr, vr = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    qr, kr = qr.reshape(-1, D).contiguous(), kr.reshape(-1, D).contiguous()
    if norm:
        qr, kr = elementwise.rms_norm(qr, qw, 1e-6), elementwise.rms_norm(kr, kw, 1e-6)
    qr = elementwise.apply_rope_(qr.view(B * S, Hq, D).clone(), pos.reshape(-1), cos, sin)
    kr = elementwise.apply_rope_(kr.view(B * S, Hkv, D).clone(), pos.reshape(-1), cos, sin)
    kc_ref, vc_ref = torch.zeros_like(kc), torch.zeros_like(vc)
    slots = (torch.arange(B, device=dev)[:, None] * max_len + pos).reshape(-1)
    elementwise.kv_append(kr, kc_ref, slots)
    elementwise.kv_append(vr.reshape(B * S, Hkv, D).contiguous(), vc_ref, slots)
    assert torch.equal(total, offs + S)
    assert torch.equal(vc, vc_ref)
    assert (q.view(B * S, Hq, D).float() - qr.float()).abs().max().item() <= 2e-2
    assert (kc.float() - kc_ref.float()).abs().max().item() <= 2e-2
    assert torch.equal(kc == 0, kc_ref == 0) or (kc != 0).sum() == (kc_ref != 0).sum()


@pytest.mark.parametrize("B,Hq,Hkv,max_len,norm,splits", [
    (1, 32, 8, 1024, True, -1), (1, 32, 8, 1024, True, 1), (4, 32, 8, 512, True, 3), (2, 8, 1, 1000, False, 2),
    (3, 64, 2, 256, True, -1), (2, 4, 4, 16, True, 1), (64, 32, 8, 2048, True, -1), (2, 32, 8, 8192, True, -1),
    (3, 16, 2, 1000, True, -1)])
def test_decode_step_from_qkv_matches_prep_then_attention(B, Hq, Hkv, max_len, norm, splits):
    """The attention kernel that norms, rotates and appends the new token itself, against the two launches it
    replaces (qkv_prep, then split-KV decode) and against an fp32 reference, at cache lengths around every block
    edge.  Unwritten cache slots hold NaN: nothing outside the attended tokens may reach the result."""
    from triton_distributed_gh_amd.ops import elementwise, flash_decode

    torch.manual_seed(B * 1000 + Hq * 10 + Hkv)
    dev, bf, D = "cuda", torch.bfloat16, 128
    W = (Hq + 2 * Hkv) * D
    qw = (torch.rand(D, device=dev) + 0.5).to(bf) if norm else None
    kw = (torch.rand(D, device=dev) + 0.5).to(bf) if norm else None
    cos, sin = elementwise.rope_cache(max_len + 8, D, device=dev)
    edge = [0, 1, 15, 16, 30, 31, 32, 33, 63, 64, 127, 128, 129, 255, 256, 511, max_len - 2, max_len - 1]
    edge = [e for e in edge if e < max_len]
    for it in range(4):
        offs = torch.tensor([edge[(it * 5 + 3 * b) % len(edge)] for b in range(B)], dtype=torch.int32, device=dev)
        pos = (offs.long() + 2)[:, None]                           # rotary position need not equal the slot
        big = (torch.rand(B, 1, W + 64, device=dev) * 2 - 1).to(bf)
        qkv = big[:, :, :W]                                        # strided rows
        kc = torch.full((B, max_len, Hkv, D), float("nan"), device=dev, dtype=bf)
        vc = torch.full_like(kc, float("nan"))
        for b in range(B):
            n = int(offs[b])
            kc[b, :n] = (torch.randn(n, Hkv, D, device=dev) * 0.5).to(bf)
            vc[b, :n] = torch.randn(n, Hkv, D, device=dev).to(bf)
        kc2, vc2 = kc.clone(), vc.clone()
        total, total2 = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
        assert flash_decode.decode_step_from_qkv_supported(qkv, kc, vc, D, cos)
        out = flash_decode.decode_step_from_qkv(qkv, qw, kw, 1e-6, 1e-6, pos, cos, sin, kc, vc, offs, total, Hq, Hkv,
                                                kv_split=splits, one_launch=bool(it & 1))
        # the two launches
        q2 = elementwise.qkv_prep(qkv, qw, kw, 1e-6, 1e-6, pos, cos, sin, kc2, vc2, offs, total2, Hq, Hkv, D)
        table = torch.arange(B, dtype=torch.int32, device=dev).view(B, 1)
        out2 = flash_decode.gqa_fwd_batch_decode_persistent(q2[:, 0], kc2, vc2, None, None, total2, table, D ** -0.5,
                                                            kv_split=splits)
        assert torch.equal(total, offs + 1) and torch.equal(total, total2)
        for b in range(B):
            n = int(offs[b])
            assert torch.equal(vc[b, :n + 1], vc2[b, :n + 1]), "V rows (old ones untouched, new one appended)"
            assert torch.equal(kc[b, :n], kc2[b, :n]), "old K rows untouched"
            assert (kc[b, n].float() - kc2[b, n].float()).abs().max().item() <= 2e-2
            assert torch.isnan(kc[b, n + 1:].float()).all() and torch.isnan(vc[b, n + 1:].float()).all(), \
                "nothing written past the new token"
        ref = flash_decode.decode_reference(q2[:, 0], kc2, vc2, total2, table, D ** -0.5)
        assert torch.isfinite(out.float()).all()
        assert (out.float() - out2.float()).abs().max().item() <= 2e-2
        assert (out.float() - ref).abs().max().item() <= 2e-2


@pytest.mark.parametrize("max_len", [64, 96, 100, 256])
def test_decode_step_from_qkv_with_a_full_cache(max_len):
    """kv_offset == max_len: the new token is attended from registers and NOT appended.  The cache tensors are the
    last allocation of their sequence (B = 1 and the table of the LAST sequence): a look-up one page too far would
    read beyond the table and the cache.  Result = attention over the max_len cached tokens plus the new one."""
    from triton_distributed_gh_amd.ops import elementwise, flash_decode

    torch.manual_seed(max_len)
    dev, bf, D, Hq, Hkv, B = "cuda", torch.bfloat16, 128, 8, 2, 2
    W = (Hq + 2 * Hkv) * D
    cos, sin = elementwise.rope_cache(max_len + 8, D, device=dev)
    qkv = (torch.rand(B, 1, W, device=dev) * 2 - 1).to(bf)
    kc = (torch.randn(B, max_len, Hkv, D, device=dev) * 0.5).to(bf)
    vc = torch.randn(B, max_len, Hkv, D, device=dev).to(bf)
    kc0, vc0 = kc.clone(), vc.clone()
    offs = torch.tensor([max_len, max_len - 1], dtype=torch.int32, device=dev)     # full, and one slot left
    pos = offs.long()[:, None]
    total = torch.zeros(B, dtype=torch.int32, device=dev)
    for one_launch in (False, True):
        kc.copy_(kc0), vc.copy_(vc0)
        out = flash_decode.decode_step_from_qkv(qkv, None, None, 1e-6, 1e-6, pos, cos, sin, kc, vc, offs, total, Hq, Hkv,
                                                one_launch=one_launch)
        torch.cuda.synchronize()
        assert total.tolist() == [max_len + 1, max_len]
        assert torch.equal(kc[0], kc0[0]) and torch.equal(vc[0], vc0[0]), "a full cache is left as it is"
        # reference: rotate q / k of the new token, attend over cached tokens + the new one
        q = qkv[:, 0, : Hq * D].view(B, Hq, D).clone()
        kn = qkv[:, 0, Hq * D:(Hq + Hkv) * D].view(B, Hkv, D).clone()
        vn = qkv[:, 0, (Hq + Hkv) * D:].view(B, Hkv, D)
        elementwise.apply_rope_(q, pos.view(-1), cos, sin)
        elementwise.apply_rope_(kn, pos.view(-1), cos, sin)
        for b in range(B):
            n = int(offs[b])
            keys = torch.cat([kc0[b, :n], kn[b:b + 1]]).float()        # [n + 1, Hkv, D]
            vals = torch.cat([vc0[b, :n], vn[b:b + 1]]).float()
            for h in range(Hq):
                g = h // (Hq // Hkv)
                p = torch.softmax(keys[:, g] @ q[b, h].float() * D ** -0.5, 0)
                ref = p @ vals[:, g]
                assert (out[b, h].float() - ref).abs().max().item() <= 2e-2, (max_len, b, h, one_launch)


def test_attention_running_max_covers_every_lane_group():
    """One key far above the rest, placed at every position of a 32-token block in turn: the running maximum
    must be the maximum over ALL lane groups of the wave.  With a maximum taken from one group only, a
    logit 400 above it overflows exp2 and the row turns to inf / nan (decode and prefill kernels share the
    cross-lane max helper)."""
    D, Hq, Hkv, page, L = 128, 8, 1, 32, 96
    dtype = torch.bfloat16
    scale = 1.0
    for pos in range(32, 64, 3):
        k = (torch.randn(4, page, Hkv, D, device="cuda") * 0.05).to(dtype)
        v = torch.randn(4, page, Hkv, D, device="cuda").to(dtype)
        q = (torch.randn(1, 1, D, device="cuda") * 0.05).to(dtype).expand(1, Hq, D).contiguous()
        table = torch.arange(4, device="cuda", dtype=torch.int32).view(1, 4)
        lens = torch.tensor([L], dtype=torch.int32, device="cuda")
        # key `pos` is parallel to the query vector (the same for every head): its logit is 400, the others ~0
        qd = q[0, 0].float()
        k.view(-1, Hkv, D)[pos, 0] = (qd / (qd ** 2).sum() * 400).to(dtype)
        ref = fd.decode_reference(q, k, v, lens, table, scale)
        out = fd.gqa_fwd_batch_decode(q, k, v, None, None, lens, table, scale, kv_split=1)
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all(), f"dominant key at {pos}: non-finite output"
        assert (out.float() - ref).abs().max().item() < 5e-2, f"dominant key at {pos}"
        # the same keys through the prefill kernel (one query row per head, no mask)
        cu_q = torch.tensor([0, 1], dtype=torch.int32, device="cuda")
        cu_k = torch.tensor([0, L], dtype=torch.int32, device="cuda")
        kk = k.view(-1, Hkv, D)[:L].contiguous()
        vv = v.view(-1, Hkv, D)[:L].contiguous()
        o2 = flash_attn_varlen_fwd(q[0].unsqueeze(0).contiguous().view(1, Hq, D), kk, vv, cu_q, cu_k, 1, causal=False,
                                   softmax_scale=scale)
        torch.cuda.synchronize()
        assert torch.isfinite(o2.float()).all(), f"prefill, dominant key at {pos}: non-finite output"
        assert (o2.float().view(1, Hq, D) - ref).abs().max().item() < 5e-2, f"prefill, dominant key at {pos}"
