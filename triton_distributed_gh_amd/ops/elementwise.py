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
