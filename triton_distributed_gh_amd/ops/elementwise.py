"""RMSNorm, RoPE, SiLU*up, add, fp8 group quantisation and KV-cache append.

GPU bf16 tensors run the vectorised HIP kernels of csrc/ops/elementwise.hip; anything else (CPU
tensors, other dtypes) takes a PyTorch path with the same semantics, so models run on both
backends.  Role in the reference: layers/amd/tp_attn.py:52-168 and the mega-kernel task kernels.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from .. import _lib


def _native(*ts) -> bool:
    return all(t is None or (t.is_cuda and t.dtype == torch.bfloat16) for t in ts)


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6, residual: Optional[torch.Tensor] = None):
    """y = rmsnorm(x [+ residual]) * weight.  With ``residual`` returns (y, x + residual)."""
    hidden = x.shape[-1]
    if _native(x, weight, residual) and hidden % 8 == 0 and x.stride(-1) == 1:
        x2 = x.reshape(-1, hidden)
        y = torch.empty_like(x2)
        res2 = residual.reshape(-1, hidden).contiguous() if residual is not None else None
        res_out = torch.empty_like(x2) if residual is not None else None
        _lib.check(
            _lib.hip().tdist_rms_norm(_lib.ptr(x2), _lib.ptr(res2), _lib.ptr(weight), _lib.ptr(y), _lib.ptr(res_out),
                                      ctypes.c_longlong(x2.shape[0]), hidden, ctypes.c_float(eps),
                                      ctypes.c_longlong(x2.stride(0)), ctypes.c_longlong(y.stride(0)),
                                      _lib.stream_handle()), "tdist_rms_norm")
        y = y.view(x.shape)
        return (y, res_out.view(x.shape)) if residual is not None else y
    xf = x.float()
    if residual is not None:
        xf = (xf + residual.float()).to(x.dtype).float()
    y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * weight.float()).to(x.dtype)
    return (y, xf.to(x.dtype)) if residual is not None else y


def rope_cache(max_pos: int, head_dim: int, theta: float = 1e6, device=None) -> Tuple[torch.Tensor, torch.Tensor]:
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32, device=device) / head_dim))
    ang = torch.outer(torch.arange(max_pos, dtype=torch.float32, device=device), inv)
    return ang.cos().contiguous(), ang.sin().contiguous()


def apply_rope_(x: torch.Tensor, positions: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """In place neox-style rotary embedding on x [tokens, heads, D]; positions [tokens]."""
    T, H, D = x.shape
    if _native(x) and D % 16 == 0 and x.stride(2) == 1 and cos.is_cuda:
        pos = positions.to(torch.int64).contiguous()
        _lib.check(
            _lib.hip().tdist_rope(_lib.ptr(x), _lib.ptr(pos), _lib.ptr(cos), _lib.ptr(sin), ctypes.c_longlong(T), H, D,
                                  ctypes.c_longlong(x.stride(0)), ctypes.c_longlong(x.stride(1)),
                                  _lib.stream_handle()), "tdist_rope")
        return x
    c = cos[positions.long()].unsqueeze(1).to(torch.float32)
    s = sin[positions.long()].unsqueeze(1).to(torch.float32)
    x1, x2 = x[..., : D // 2].float(), x[..., D // 2:].float()
    out = torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1).to(x.dtype)
    x.copy_(out)
    return x


def qkv_prep_supported(qkv: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, head_dim: int) -> bool:
    return (_native(qkv, k_cache, v_cache) and head_dim in (64, 128, 256) and qkv.dim() == 3 and qkv.stride(2) == 1
            and qkv.stride(1) % 8 == 0 and (qkv.shape[1] == 1 or qkv.stride(0) == qkv.shape[1] * qkv.stride(1))
            and k_cache.is_contiguous() and v_cache.is_contiguous() and qkv.data_ptr() % 16 == 0)


def qkv_prep(qkv: torch.Tensor, q_w: Optional[torch.Tensor], k_w: Optional[torch.Tensor], q_eps: float, k_eps: float,
             position_ids: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, k_cache: torch.Tensor,
             v_cache: torch.Tensor, kv_offset: torch.Tensor, total_out: torch.Tensor, num_q_heads: int,
             num_kv_heads: int, head_dim: int) -> torch.Tensor:
    """One launch for everything between the QKV projection and attention.

    ``qkv [B, S, (Hq + 2 Hkv) * D]`` (row stride free): q/k heads get their optional per-head RMSNorm
    (``q_w`` / ``k_w`` [D]) and the neox rotation at ``position_ids [B, S]``; k and v rows are appended
    to ``k_cache / v_cache [B, max_len, Hkv, D]`` at ``kv_offset[b] + s`` (int32 [B]); ``total_out[b]``
    (int32 [B]) receives ``kv_offset[b] + S``.  Returns q ``[B, S, Hq, D]``.  Call
    ``qkv_prep_supported`` first."""
    B, S, _ = qkv.shape
    q = torch.empty((B, S, num_q_heads, head_dim), dtype=qkv.dtype, device=qkv.device)
    pos = position_ids.to(torch.int64).contiguous()
    assert kv_offset.dtype == torch.int32 and total_out.dtype == torch.int32 and cos.dtype == torch.float32
    _lib.check(
        _lib.hip().tdist_qkv_prep(_lib.ptr(qkv), ctypes.c_longlong(qkv.stride(1)), _lib.ptr(q_w), _lib.ptr(k_w),
                                  ctypes.c_float(q_eps), ctypes.c_float(k_eps), _lib.ptr(pos), _lib.ptr(cos),
                                  _lib.ptr(sin), _lib.ptr(q), _lib.ptr(k_cache), _lib.ptr(v_cache), _lib.ptr(kv_offset),
                                  _lib.ptr(total_out), ctypes.c_longlong(B * S), S, num_q_heads, num_kv_heads, head_dim,
                                  ctypes.c_longlong(k_cache.shape[1]), _lib.stream_handle()), "tdist_qkv_prep")
    return q


def silu_mul(gate_up: torch.Tensor) -> torch.Tensor:
    """gate_up [..., 2*I] -> silu(gate) * up  [..., I]."""
    inter = gate_up.shape[-1] // 2
    if _native(gate_up) and inter % 8 == 0 and gate_up.is_contiguous():
        g2 = gate_up.view(-1, 2 * inter)
        y = torch.empty((g2.shape[0], inter), dtype=gate_up.dtype, device=gate_up.device)
        _lib.check(_lib.hip().tdist_silu_mul(_lib.ptr(g2), _lib.ptr(y), ctypes.c_longlong(g2.shape[0]), inter,
                                             _lib.stream_handle()), "tdist_silu_mul")
        return y.view(*gate_up.shape[:-1], inter)
    g, u = gate_up[..., :inter], gate_up[..., inter:]
    return (torch.nn.functional.silu(g.float()) * u.float()).to(gate_up.dtype)


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    if _native(a, b) and a.is_contiguous() and b.is_contiguous() and a.numel() % 8 == 0 and a.shape == b.shape:
        y = torch.empty_like(a)
        _lib.check(_lib.hip().tdist_add(_lib.ptr(a), _lib.ptr(b), _lib.ptr(y), ctypes.c_longlong(a.numel()),
                                        _lib.stream_handle()), "tdist_add")
        return y
    return a + b


def fp8_group_quant(x: torch.Tensor, group: int = 128) -> Tuple[torch.Tensor, torch.Tensor]:
    """bf16 [M, H] -> (e4m3fn [M, H], f32 scale [M, H/128])."""
    M, H = x.shape
    assert group == 128 and H % 128 == 0 and x.is_contiguous() and x.dtype == torch.bfloat16 and x.is_cuda
    q = torch.empty((M, H), dtype=torch.float8_e4m3fn, device=x.device)
    scale = torch.empty((M, H // 128), dtype=torch.float32, device=x.device)
    _lib.check(_lib.hip().tdist_fp8_group_quant(_lib.ptr(x), _lib.ptr(q), _lib.ptr(scale),
                                                ctypes.c_longlong(M * (H // 128)), _lib.stream_handle()),
               "tdist_fp8_group_quant")
    return q, scale


def kv_append(src: torch.Tensor, cache: torch.Tensor, slots: torch.Tensor) -> None:
    """cache.view(-1, row)[slots[t]] = src[t]  (src [T, kv_heads, D], cache [..., kv_heads, D])."""
    T = src.shape[0]
    row = src[0].numel()
    if _native(src, cache) and row % 8 == 0 and src.stride(-1) == 1 and cache.is_contiguous() and src[0].is_contiguous():
        s = slots.to(torch.int64).contiguous()
        _lib.check(_lib.hip().tdist_kv_append(_lib.ptr(src), _lib.ptr(cache), _lib.ptr(s), ctypes.c_longlong(T), row,
                                              ctypes.c_longlong(src.stride(0)), _lib.stream_handle()),
                   "tdist_kv_append")
        return
    cache.view(-1, row)[slots.long()] = src.reshape(T, row)
