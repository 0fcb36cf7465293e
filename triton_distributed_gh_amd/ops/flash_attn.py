"""Flash-attention forward (varlen, causal, GQA) and the kv-cache attention used by the TP layers.

``flash_attn_varlen_func`` / ``flash_attn_with_kvcache`` cover what the reference's AMD attention
layer takes from the external flash_attn package (layers/amd/tp_attn.py:275-308); both run the
gfx950 kernels in csrc/ops/flash_attn.hip and csrc/ops/flash_decode.hip.  CPU tensors take a
PyTorch fp32 path so models also run without a GPU.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch

from .. import _lib
from . import elementwise, flash_decode


def attention_reference(q, k, v, cu_q, cu_k, causal=True, scale=None, q_positions=None):
    """fp32 PyTorch oracle.  q [Tq, Hq, D], k/v [Tk, Hkv, D]; ``q_positions`` (optional, [Tq])
    gives each query row's position inside its sequence (default: the last Lq positions)."""
    Tq, Hq, D = q.shape
    Hkv = k.shape[1]
    scale = scale if scale is not None else 1.0 / math.sqrt(D)
    out = torch.zeros(Tq, Hq, v.shape[-1], dtype=torch.float32, device=q.device)
    cu_q, cu_k = [int(x) for x in cu_q], [int(x) for x in cu_k]
    for b in range(len(cu_q) - 1):
        qs, qe, ks, ke = cu_q[b], cu_q[b + 1], cu_k[b], cu_k[b + 1]
        Lq, Lk = qe - qs, ke - ks
        if Lq == 0 or Lk == 0:
            continue
        qq = q[qs:qe].float().view(Lq, Hkv, Hq // Hkv, D)
        s = torch.einsum("qhgd,khd->hgqk", qq, k[ks:ke].float()) * scale
        if causal:
            pos = q_positions[qs:qe].to(q.device) if q_positions is not None else \
                torch.arange(Lq, device=q.device) + (Lk - Lq)
            mask = torch.arange(Lk, device=q.device)[None, :] <= pos[:, None]
            s = s.masked_fill(~mask, float("-inf"))
        p = torch.softmax(s, -1).nan_to_num(0.0)
        out[qs:qe] = torch.einsum("hgqk,khd->qhgd", p, v[ks:ke].float()).reshape(Lq, Hq, -1)
    return out


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return 0
    if t.dtype == torch.float16:
        return 1
    raise TypeError(f"attention supports bf16 / fp16, got {t.dtype}")


def flash_attn_varlen_fwd(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, out=None, softmax_scale=None,
                          causal=True, pos_mode=0, rank=0, world_size=1, flags=None, epoch=0, err=None,
                          return_lse=False):
    """Native varlen attention.  ``pos_mode`` 0: query rows are the last Lq positions of their
    sequence; 1 / 2: rows are rank ``rank``'s contiguous / zig-zag shard of a sequence split over
    ``world_size`` ranks.  ``flags`` (int32 [world_size]) makes the kernel wait for a source rank's
    KV slab (== ``epoch``) before reading it."""
    Tq, Hq, D = q.shape
    Hkv = k.shape[1]
    assert q.is_cuda and k.dtype == q.dtype == v.dtype and v.shape[-1] == D and k.shape[-1] == D
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    assert cu_seqlens_q.dtype == torch.int32 and cu_seqlens_k.dtype == torch.int32
    assert cu_seqlens_q.numel() == cu_seqlens_k.numel()
    batch = cu_seqlens_q.numel() - 1
    out = out if out is not None else torch.empty_like(q)
    assert out.stride(2) == 1 and out.shape == q.shape
    lse = torch.empty((Hq, Tq), dtype=torch.float32, device=q.device) if return_lse else None
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    ll = ctypes.c_longlong
    _lib.check(
        _lib.hip().tdist_flash_attn_varlen_fwd(
            _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(out), _lib.ptr(lse), _lib.ptr(cu_seqlens_q),
            _lib.ptr(cu_seqlens_k), batch, int(max_seqlen_q), Hq, Hkv, D, ll(q.stride(0)), ll(q.stride(1)),
            ll(k.stride(0)), ll(k.stride(1)), ll(v.stride(0)), ll(v.stride(1)), ll(out.stride(0)), ll(out.stride(1)),
            ll(Tq), ctypes.c_float(scale), 1 if causal else 0, pos_mode, rank, world_size, _lib.ptr(flags), epoch,
            _lib.ptr(err), _dt(q), _lib.stream_handle()), "tdist_flash_attn_varlen_fwd")
    return (out, lse) if return_lse else out


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k=None, softmax_scale=None,
                           causal=True):
    """flash_attn-compatible entry point (bottom-right aligned causal mask)."""
    if q.is_cuda and q.dtype in (torch.bfloat16, torch.float16) and q.shape[-1] in (64, 128):
        return flash_attn_varlen_fwd(q, k, v, cu_seqlens_q.int(), cu_seqlens_k.int(), max_seqlen_q,
                                     softmax_scale=softmax_scale, causal=causal)
    return attention_reference(q, k, v, cu_seqlens_q.tolist(), cu_seqlens_k.tolist(), causal, softmax_scale).to(q.dtype)


def flash_attn_func(q, k, v, softmax_scale=None, causal=True):
    """q [B, Sq, Hq, D], k/v [B, Sk, Hkv, D] -> [B, Sq, Hq, D]."""
    B, Sq, Hq, D = q.shape
    Sk = k.shape[1]
    cu_q = torch.arange(0, (B + 1) * Sq, Sq, dtype=torch.int32, device=q.device)
    cu_k = torch.arange(0, (B + 1) * Sk, Sk, dtype=torch.int32, device=q.device)
    o = flash_attn_varlen_func(q.reshape(B * Sq, Hq, D), k.reshape(B * Sk, -1, D), v.reshape(B * Sk, -1, D), cu_q, cu_k,
                               Sq, Sk, softmax_scale, causal)
    return o.view(B, Sq, Hq, D)


_ARANGE = {}


def _identity_table(device, bsz: int, pages: int) -> torch.Tensor:
    """[bsz, pages] page table 0..bsz*pages-1.  Cached per stream: the fill is ordered only against
    the stream that created it, and creating it must not force a host synchronisation."""
    key = (device, bsz, pages, torch.cuda.current_stream(device).cuda_stream)
    t = _ARANGE.get(key)
    if t is None:
        t = torch.arange(bsz * pages, dtype=torch.int32, device=device).view(bsz, pages)
        if not torch.cuda.is_current_stream_capturing():
            _ARANGE[key] = t
    return t


def flash_attn_with_kvcache(q, k_cache, v_cache, k=None, v=None, cache_seqlens=None, softmax_scale=None, causal=True,
                            seq_table=None):
    """q [B, Sq, Hq, D]; caches [B, max_len, Hkv, D] (contiguous per sequence); ``k``/``v``
    [B, Sq, Hkv, D] are appended at ``cache_seqlens`` first.  Sq == 1 runs split-KV decode, longer
    queries the prefill kernel over the cache prefix."""
    B, Sq, Hq, D = q.shape
    max_len, Hkv = k_cache.shape[1], k_cache.shape[2]
    if cache_seqlens is None:
        cache_seqlens = torch.zeros(B, dtype=torch.int32, device=q.device)
    if cache_seqlens.dtype != torch.int32:
        cache_seqlens = cache_seqlens.to(torch.int32)
    if k is not None:
        pos = cache_seqlens.long().unsqueeze(1) + torch.arange(Sq, device=q.device).unsqueeze(0)   # [B, Sq]
        slots = (torch.arange(B, device=q.device).unsqueeze(1) * max_len + pos).reshape(-1)
        elementwise.kv_append(k.reshape(B * Sq, Hkv, D), k_cache, slots)
        elementwise.kv_append(v.reshape(B * Sq, Hkv, D), v_cache, slots)
    # with k / v already in the cache, ``cache_seqlens`` is the length to attend over and ``seq_table``
    # (int32 [B, 1] = arange) may be passed in, so a decode step launches nothing but the attention kernel
    total = (cache_seqlens + Sq).to(torch.int32) if k is not None else cache_seqlens
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    native = q.is_cuda and q.dtype in (torch.bfloat16, torch.float16) and k_cache.is_contiguous() and v_cache.is_contiguous()
    if Sq == 1 and native and D in (64, 128, 256):
        table = seq_table if seq_table is not None else _identity_table(q.device, B, 1)
        # one "page" per sequence: page size = max_len
        kc = k_cache.view(B, max_len, Hkv, D)
        vc = v_cache.view(B, max_len, Hkv, D)
        # two launches (split, head-parallel merge): ahead of the single-launch form inside a HIP graph
        o = flash_decode.gqa_fwd_batch_decode(q[:, 0], kc, vc, None, None, total, table, scale)
        return o.view(B, 1, Hq, D)
    if native and D in (64, 128):
        # the cache rows of sequence b live at b * max_len + [0, total_b): express as varlen over the cache
        cu_q = torch.arange(0, (B + 1) * Sq, Sq, dtype=torch.int32, device=q.device)
        if B == 1:
            cu_k = torch.stack([torch.zeros((), dtype=torch.int32, device=q.device), total[0]]).int()
            o = flash_attn_varlen_fwd(q[0], k_cache.view(-1, Hkv, D), v_cache.view(-1, Hkv, D), cu_q, cu_k,
                                      Sq, softmax_scale=scale, causal=causal)
            return o.view(1, Sq, Hq, D)
        outs = []
        for b in range(B):
            cu_k = torch.stack([torch.zeros((), dtype=torch.int32, device=q.device), total[b]]).int()
            outs.append(flash_attn_varlen_fwd(q[b], k_cache[b], v_cache[b], cu_q[:2], cu_k, Sq, softmax_scale=scale,
                                              causal=causal))
        return torch.stack(outs)
    outs = []
    for b in range(B):
        L = int(total[b])
        o = attention_reference(q[b], k_cache[b, :L], v_cache[b, :L], [0, Sq], [0, L], causal, scale)
        outs.append(o.to(q.dtype))
    return torch.stack(outs)
