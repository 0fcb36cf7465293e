"""GQA decode attention over a paged KV cache (split-KV + LSE merge) and its distributed merges.

Host API of the reference's kernels/nvidia/flash_decode.py: ``gqa_fwd_batch_decode`` (:763-844),
``gqa_fwd_batch_decode_intra_rank`` (:847-928, also returns the LSE column so ranks can be merged),
``gqa_fwd_batch_decode_persistent`` (:931-976, one launch) and the inter-rank combine (:452-532).
All of them run csrc/ops/flash_decode.hip on the GPU; CPU tensors take the PyTorch reference path.

Differences by design: partial results are always fp32 (the reference keeps them in q's dtype),
the "persistent" variant merges in the last workgroup to finish instead of a grid barrier, and the
default split count is sized for 256 CUs rather than fixed at 32.
"""
from __future__ import annotations

import ctypes
import os
import math
from typing import Optional

import torch

from .. import _lib

_TICKETS = {}
ONE_LAUNCH = os.environ.get("TDIST_DECODE_ONE_LAUNCH", "0") != "0"      # decode_step_from_qkv: in-kernel merge


def _tickets(device, n: int) -> torch.Tensor:
    """Arrival counters of the fused merge (self-resetting).  One array per (device, stream): two
    decode launches that run concurrently on different streams must not share counters."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _TICKETS.get(key)
    if t is None or t.numel() < n:
        t = torch.zeros(max(n, 4096), dtype=torch.int32, device=device)
        _TICKETS[key] = t
    return t


def auto_kv_splits(batch: int, q_heads: int, kv_heads: int, max_kv_len: int, num_cus: int = 256,
                   fused: bool = False, page_size: int = 0) -> int:
    """Number of KV splits of a decode launch (the grid; with an automatic count every sequence then uses as many
    of them as its own length calls for, see ``DecodeArgs::adaptive``).

    A workgroup holds a CU by itself at head_dim 128 (232 - 256 VGPRs), so the workgroups should fit ONE round of
    the chip: at most ``num_cus / (batch * kv_heads * head_blocks)`` splits.  Inside that bound, measured on
    MI355X with cold K / V (scripts/dev/probe_decode_splits.py, 32 / 8 heads, HIP graph):

    * two launches (split kernel, then a merge kernel that is parallel over heads): s ~ sqrt(L / 16), at least 128
      tokens per split, at most 32 splits - batch 1: 15.2 us at 2048 tokens (16 splits; rule: 11), 18.1 at 4096
      (16; 20.2 at 32), 23.3 at 8192 (16; 24.5 at 32; rule: 23), 43.6 at 32768 (32) - with the merge launch that
      loads all rows of a head in one round trip these became 17.0 at 8192 and 36.5 at 32768;
    * ``fused`` (one launch; the last workgroup of a (sequence, head block) merges its splits, one wave per head
      with every row in flight at once): the same rule - batch 1: 11.7 us at 2048 (16 splits), 17.9 at 8192 (24),
      36.6 at 32768 (32), 109.9 at 131072 (32); level with the two-launch form at every measured shape.

    Up to two rounds of workgroups are allowed (batch 16 x 4096: 60.6 us with 4 splits = 512 workgroups against 68.5
    with 2; batch 4 x 8192: 33.8 with 16 against 35.1 with 8).

    Pages shorter than a 32-token block (``page_size`` 1 - 16) take the same counts: the kernel stages its split's
    slice of the page table in LDS, which brought them within 3 % of whole-sequence pages (batch 1, 131072 tokens:
    115 us at page 16, 119 at page 1, 112 with one page per sequence; 245 / 253 us before)."""
    group = q_heads // kv_heads
    base = batch * kv_heads * ((group + 15) // 16)
    # two rounds' worth of workgroups at most: the latency form of the kernel (150 registers) keeps two per CU resident
    bound = max(1, (2 * num_cus) // base)
    want = max(1, int(round(math.sqrt(max_kv_len / 16.0))))
    if fused:
        return int(max(1, min(want, bound, max(1, max_kv_len // 64), 32)))
    return int(max(1, min(want, bound, max(1, max_kv_len // 128), 32)))


def _dt(q: torch.Tensor) -> int:
    if q.dtype == torch.bfloat16:
        return 0
    if q.dtype == torch.float16:
        return 1
    raise TypeError(f"flash decode supports bf16 / fp16, got {q.dtype}")


def decode_reference(q, k_cache, v_cache, kv_lens, block_table, scale, soft_cap=0.0, return_lse=False):
    """Plain PyTorch fp32 decode attention over the paged cache (test oracle and CPU path)."""
    batch, q_heads, D = q.shape
    _, page, kv_heads, _ = k_cache.shape
    g = q_heads // kv_heads
    out = torch.zeros(batch, q_heads, v_cache.shape[-1], dtype=torch.float32, device=q.device)
    lse = torch.full((batch, q_heads), float("-inf"), dtype=torch.float32, device=q.device)
    for b in range(batch):
        L = int(kv_lens[b])
        if L == 0:
            continue
        npg = (L + page - 1) // page
        pages = block_table[b, :npg].long()
        k = k_cache[pages].reshape(-1, kv_heads, D)[:L].float()
        v = v_cache[pages].reshape(-1, kv_heads, v_cache.shape[-1])[:L].float()
        qh = q[b].float().view(kv_heads, g, D)
        s = torch.einsum("hgd,lhd->hgl", qh, k) * scale
        if soft_cap > 0:
            s = soft_cap * torch.tanh(s / soft_cap)
        lse[b] = torch.logsumexp(s, -1).reshape(-1)
        out[b] = torch.einsum("hgl,lhd->hgd", torch.softmax(s, -1), v).reshape(q_heads, -1)
    return (out, lse) if return_lse else out


def _launch(q, k_cache, v_cache, kv_lens, block_table, scale, soft_cap, splits, fused, partial, out):
    batch, q_heads, D = q.shape
    _, page, kv_heads, _ = k_cache.shape
    assert v_cache.shape[-1] == D, "the native kernel needs equal K and V head dims"
    assert q.stride(2) == 1 and k_cache.stride(3) == 1 and v_cache.stride(3) == 1
    assert k_cache.stride(0) == page * k_cache.stride(1) and v_cache.stride(0) == page * v_cache.stride(1), \
        "pages must be contiguous runs of slots"
    assert kv_lens.dtype == torch.int32 and block_table.dtype == torch.int32 and block_table.stride(1) == 1
    assert kv_lens.numel() >= batch and block_table.shape[0] >= batch
    assert partial.dtype == torch.float32 and partial.is_contiguous() and partial.numel() >= batch * q_heads * splits * (D + 1)
    group = q_heads // kv_heads
    tick = _tickets(q.device, batch * kv_heads * ((group + 15) // 16)) if fused else None
    _lib.check(
        _lib.hip().tdist_flash_decode_split(
            _lib.ptr(q), _lib.ptr(k_cache), _lib.ptr(v_cache), _lib.ptr(block_table), _lib.ptr(kv_lens),
            _lib.ptr(partial), _lib.ptr(out), _lib.ptr(tick), batch, q_heads, kv_heads, D, splits, page,
            block_table.stride(0), ctypes.c_longlong(q.stride(0)), ctypes.c_longlong(q.stride(1)),
            ctypes.c_longlong(k_cache.stride(1)), ctypes.c_longlong(k_cache.stride(2)),
            ctypes.c_longlong(v_cache.stride(1)), ctypes.c_longlong(v_cache.stride(2)),
            ctypes.c_longlong(out.stride(0) if out is not None else 0),
            ctypes.c_longlong(out.stride(1) if out is not None else 0), ctypes.c_float(scale),
            ctypes.c_float(soft_cap), fused, _dt(q), _lib.stream_handle()), "tdist_flash_decode_split")


def _combine(partial, strides, n, batch, heads, D, lens, out, out_mode, dtype_code):
    _lib.check(
        _lib.hip().tdist_flash_decode_combine(
            _lib.ptr(partial), ctypes.c_longlong(strides[0]), ctypes.c_longlong(strides[1]),
            ctypes.c_longlong(strides[2]), n, batch, heads, D, _lib.ptr(lens), lens.stride(0) if lens is not None else 0,
            _lib.ptr(out), ctypes.c_longlong(out.stride(0)), ctypes.c_longlong(out.stride(1)), out_mode, dtype_code,
            _lib.stream_handle()), "tdist_flash_decode_combine")


def _splits(q, k_cache, block_table, kv_split, fused=False):
    if kv_split is not None and kv_split > 0:
        return int(kv_split)
    forced = int(os.environ.get("TDIST_DECODE_KV_SPLITS", "0"))        # tuning knob
    if forced > 0:
        return forced
    batch, q_heads, _ = q.shape
    return auto_kv_splits(batch, q_heads, k_cache.shape[2], block_table.shape[1] * k_cache.shape[1], fused=fused,
                          page_size=k_cache.shape[1])


def _splits_forced(kv_split) -> bool:
    return bool(kv_split is not None and kv_split > 0) or int(os.environ.get("TDIST_DECODE_KV_SPLITS", "0")) > 0


def _partial(q, splits, output_split):
    batch, q_heads, D = q.shape
    if output_split is not None and output_split.dtype == torch.float32 and output_split.is_contiguous() \
            and output_split.numel() >= batch * q_heads * splits * (D + 1):
        return output_split
    return torch.empty((batch, q_heads, splits, D + 1), dtype=torch.float32, device=q.device)


def gqa_fwd_batch_decode(q, k_cache, v_cache, workspace, q_lens, kv_lens, block_table, scale, soft_cap=0.0,
                         output_split=None, output_combine=None, kv_split=-1):
    """q [batch, q_heads, D]; k/v_cache [pages, page_size, kv_heads, D]; returns [batch, q_heads, D]
    in q's dtype.  Two launches: split-KV then merge (reference :763-844)."""
    batch, q_heads, D = q.shape
    if not q.is_cuda:
        return decode_reference(q, k_cache, v_cache, kv_lens, block_table, scale, soft_cap).to(q.dtype)
    splits = _splits(q, k_cache, block_table, kv_split)
    part = _partial(q, splits, output_split)
    out = output_combine if (output_combine is not None and output_combine.dtype == q.dtype) else \
        torch.empty((batch, q_heads, D), dtype=q.dtype, device=q.device)
    _launch(q, k_cache, v_cache, kv_lens, block_table, scale, soft_cap, splits, 0 if _splits_forced(kv_split) else 4, part, None)
    _combine(part, (D + 1, q_heads * splits * (D + 1), splits * (D + 1)), splits, batch, q_heads, D, None, out, 1, _dt(q))
    return out


def gqa_fwd_batch_decode_intra_rank(q, k_cache, v_cache, workspace, q_lens, kv_lens, block_table, scale, soft_cap=0.0,
                                    output_split=None, output_combine=None, kv_split=-1):
    """Like ``gqa_fwd_batch_decode`` but returns fp32 [batch, q_heads, D + 1] with the natural-log
    LSE in the last column, the form the inter-rank merge consumes (reference :847-928)."""
    batch, q_heads, D = q.shape
    if not q.is_cuda:
        o, lse = decode_reference(q, k_cache, v_cache, kv_lens, block_table, scale, soft_cap, return_lse=True)
        return torch.cat([o, lse.unsqueeze(-1)], -1)
    splits = _splits(q, k_cache, block_table, kv_split)
    part = _partial(q, splits, output_split)
    out = output_combine if (output_combine is not None and output_combine.dtype == torch.float32) else \
        torch.empty((batch, q_heads, D + 1), dtype=torch.float32, device=q.device)
    # two launches: at small batch the split workgroups are so short that the in-kernel hand-off
    # of the fused form (drain + ticket per workgroup) costs more than the kernel boundary it
    # saves (batch 1, 32k tokens: 40 us vs 65 us on MI355X)
    _launch(q, k_cache, v_cache, kv_lens, block_table, scale, soft_cap, splits, 0 if _splits_forced(kv_split) else 4, part, None)
    _combine(part, (D + 1, q_heads * splits * (D + 1), splits * (D + 1)), splits, batch, q_heads, D, None, out, 2, _dt(q))
    return out


def gqa_fwd_batch_decode_persistent(q, k_cache, v_cache, workspace, q_lens, kv_lens, block_table, scale, soft_cap=0.0,
                                    output_split=None, output_combine=None, kv_split=-1):
    """Single-launch decode: the last workgroup of every (batch, head block) merges the splits
    (reference :931-976 uses a persistent grid with a grid barrier)."""
    batch, q_heads, D = q.shape
    if not q.is_cuda:
        return decode_reference(q, k_cache, v_cache, kv_lens, block_table, scale, soft_cap).to(q.dtype)
    splits = _splits(q, k_cache, block_table, kv_split, fused=True)
    part = _partial(q, splits, output_split)
    out = output_combine if (output_combine is not None and output_combine.dtype == q.dtype) else \
        torch.empty((batch, q_heads, D), dtype=q.dtype, device=q.device)
    # an automatic split count comes from the cache CAPACITY: let every sequence use what its own length needs (+ 4)
    _launch(q, k_cache, v_cache, kv_lens, block_table, scale, soft_cap, splits, 1 + (0 if _splits_forced(kv_split) else 4),
            part, out)
    return out


def decode_step_from_qkv_supported(qkv: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, head_dim: int,
                                   cos: torch.Tensor) -> bool:
    """Shapes ``decode_step_from_qkv`` is built for: bf16, head_dim 128, one new token per sequence, contiguous
    per-sequence caches ``[B, max_len, Hkv, 128]``, fp32 rotation tables on the GPU."""
    return (qkv.is_cuda and qkv.dtype == torch.bfloat16 and k_cache.dtype == v_cache.dtype == torch.bfloat16
            and head_dim == 128 and qkv.dim() == 3 and qkv.shape[1] == 1 and qkv.stride(2) == 1 and qkv.stride(0) % 8 == 0
            and qkv.data_ptr() % 16 == 0 and k_cache.dim() == 4 and k_cache.shape[0] == qkv.shape[0]
            and k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape == v_cache.shape
            and cos.is_cuda and cos.dtype == torch.float32 and cos.is_contiguous() and cos.shape[-1] == 64
            and os.environ.get("TDIST_DECODE_FUSED_PREP", "1") != "0")


def decode_step_from_qkv(qkv: torch.Tensor, q_w: Optional[torch.Tensor], k_w: Optional[torch.Tensor], q_eps: float,
                         k_eps: float, position_ids: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                         k_cache: torch.Tensor, v_cache: torch.Tensor, kv_offset: torch.Tensor, total_out: torch.Tensor,
                         num_q_heads: int, num_kv_heads: int, seq_table: Optional[torch.Tensor] = None,
                         scale: Optional[float] = None, kv_split: int = -1,
                         one_launch: Optional[bool] = None) -> torch.Tensor:
    """From the QKV projection of a decode step to the attention output without a launch for the projection's epilogue.

    ``qkv [B, 1, (Hq + 2 Hkv) * 128]``: q / k heads get their optional per-head RMSNorm and the neox rotation at
    ``position_ids [B, 1]``, K / V of the new token go to slot ``kv_offset[b]`` of ``k_cache / v_cache
    [B, max_len, Hkv, 128]``, ``total_out[b] = kv_offset[b] + 1`` and the token attends over those tokens.
    Returns ``[B, Hq, 128]``.  The same arithmetic as ``elementwise.qkv_prep`` followed by
    ``gqa_fwd_batch_decode_persistent`` - the workgroups of the attention kernel prepare their own heads
    (csrc/include/tdist/flash_decode_device.hpp, PREP) - without the launch in between.  The splits are merged by
    a second, head-parallel launch (default; ahead of the in-kernel merge at every measured shape once the step
    sits in a HIP graph) or by the last workgroup of each head block (``one_launch`` / TDIST_DECODE_ONE_LAUNCH=1).
    Call ``decode_step_from_qkv_supported`` first."""
    B, D = qkv.shape[0], 128
    max_len = k_cache.shape[1]
    assert kv_offset.dtype == torch.int32 and total_out.dtype == torch.int32 and kv_offset.numel() >= B
    assert qkv.shape[2] == (num_q_heads + 2 * num_kv_heads) * D and k_cache.shape[2] == num_kv_heads
    pos = position_ids.reshape(-1)
    if pos.dtype != torch.int64 or not pos.is_contiguous():
        pos = pos.to(torch.int64).contiguous()
    assert pos.numel() == B
    if seq_table is None:
        seq_table = torch.arange(B, dtype=torch.int32, device=qkv.device).view(B, 1)
    assert seq_table.dtype == torch.int32 and seq_table.numel() >= B
    for w in (q_w, k_w):
        assert w is None or (w.dtype == torch.bfloat16 and w.numel() == D and w.is_contiguous())
    fused = ONE_LAUNCH if one_launch is None else bool(one_launch)
    splits = int(kv_split) if kv_split and kv_split > 0 else \
        int(os.environ.get("TDIST_DECODE_KV_SPLITS", "0")) or auto_kv_splits(B, num_q_heads, num_kv_heads, max_len, fused=fused)
    part = torch.empty((B, num_q_heads, splits, D + 1), dtype=torch.float32, device=qkv.device)
    out = torch.empty((B, num_q_heads, D), dtype=qkv.dtype, device=qkv.device)
    tick = _tickets(qkv.device, B * num_kv_heads * ((num_q_heads // num_kv_heads + 15) // 16)) if fused else None
    _lib.check(
        _lib.hip().tdist_flash_decode_prep(
            _lib.ptr(qkv), ctypes.c_longlong(qkv.stride(0)), _lib.ptr(q_w), _lib.ptr(k_w), ctypes.c_float(q_eps),
            ctypes.c_float(k_eps), _lib.ptr(pos), _lib.ptr(cos), _lib.ptr(sin), _lib.ptr(k_cache), _lib.ptr(v_cache),
            _lib.ptr(kv_offset), _lib.ptr(total_out), _lib.ptr(seq_table), _lib.ptr(part), _lib.ptr(out), _lib.ptr(tick),
            B, num_q_heads, num_kv_heads, ctypes.c_longlong(max_len), splits, ctypes.c_longlong(out.stride(0)),
            ctypes.c_longlong(out.stride(1)), ctypes.c_float(scale if scale is not None else 1.0 / math.sqrt(D)),
            0 if _splits_forced(kv_split) else 1, 1 if fused else 0, _lib.stream_handle()), "tdist_flash_decode_prep")
    if not fused:
        _combine(part, (D + 1, num_q_heads * splits * (D + 1), splits * (D + 1)), splits, B, num_q_heads, D, None, out, 1, 0)
    return out


# the reference exposes AOT-compiled twins of the two entry points; here every kernel is AOT
gqa_fwd_batch_decode_aot = gqa_fwd_batch_decode
gqa_fwd_batch_decode_intra_rank_aot = gqa_fwd_batch_decode_intra_rank


def gqa_fwd_batch_decode_inter_rank_combine(partials: torch.Tensor, kv_lens_per_rank: Optional[torch.Tensor],
                                            out: Optional[torch.Tensor] = None, dtype=torch.bfloat16):
    """partials fp32 [num_ranks, batch, q_heads, D + 1] (LSE last) -> [batch, q_heads, D] in
    ``dtype``; ranks whose ``kv_lens_per_rank [num_ranks, batch]`` entry is 0 are skipped
    (reference :452-532)."""
    R, batch, heads, D1 = partials.shape
    D = D1 - 1
    if out is None:
        out = torch.empty((batch, heads, D), dtype=dtype, device=partials.device)
    if not partials.is_cuda:
        lse = partials[..., D].clone()
        if kv_lens_per_rank is not None:
            lse = lse.masked_fill((kv_lens_per_rank <= 0).unsqueeze(-1), float("-inf"))
        w = torch.softmax(lse, 0).nan_to_num(0.0)
        out.copy_((w.unsqueeze(-1) * partials[..., :D]).sum(0).to(out.dtype))
        return out
    assert partials.dtype == torch.float32 and partials.is_contiguous()
    lens = kv_lens_per_rank
    if lens is not None:
        assert lens.dtype == torch.int32 and lens.shape == (R, batch) and lens.is_contiguous()
    _combine(partials, (batch * heads * D1, heads * D1, D1), R, batch, heads, D, lens, out, 1,
             0 if out.dtype == torch.bfloat16 else 1)
    return out


def default_scale(head_dim: int) -> float:
    return 1.0 / math.sqrt(head_dim)
