"""Low-latency AllGather + GEMM and GEMM + ReduceScatter for decode-sized problems (csrc/ops/tp_ll.hip).

``ag_gemm_intra_node`` / ``gemm_rs_intra_node`` route here when the whole problem has at most ``LL_MAX_M``
rows: the kernels run the mid-M tile body (weights streamed once, split-K across workgroups) instead of the
256x256 tile, handle the small collective part whole, and number their calls on the device - so the launch
is the same every time and may be captured into a HIP graph.  That is what lets ``Engine(backend=
"triton_dist")`` decode from a graph at world size > 1, as the reference does
(reference: python/triton_dist/models/engine.py:75-105, kernels/amd/all_gather_gemm.py:310-312).
"""
from __future__ import annotations

import ctypes
import os

import torch

from .. import _lib
from .gemm import _DT

# rows (summed over the ranks) up to which the low-latency forms run; 0 switches them off
LL_MAX_M = int(os.environ.get("TDIST_TP_LL_MAX_M", "256"))


def create_ll_state(env, num_ranks: int, max_M: int, N: int, K: int, dtype, kind: str):
    """Buffers of one context: ``kind`` "ag" (gathered workspace [rows, K] per parity) or "rs" (scatter
    buffers [w, rows / w, N] per parity); flags, split-K workspace, counters, call state, error word.
    Returns None when the low-latency form does not apply (CPU back-end, switched off)."""
    if env.backend != "hip" or num_ranks < 2 or LL_MAX_M <= 0:
        return None
    rows = min(max_M, LL_MAX_M)
    rows -= rows % num_ranks
    if rows <= 0:
        return None
    heap = env.heap
    pb, nc = ctypes.c_longlong(), ctypes.c_int()
    _lib.check(_lib.hip().tdist_tp_ll_sizes(rows, N, ctypes.byref(pb), ctypes.byref(nc)), "tdist_tp_ll_sizes")
    st = dict(kind=kind, rows=rows, data=[], flags=[])
    for _ in range(2):
        st["data"].append(heap.create_tensor_list([rows, K if kind == "ag" else N], dtype, zero=False))
        st["flags"].append(heap.create_tensor_list([max(num_ranks, 4)], torch.int32, arena="sig"))
    st["partial"] = torch.empty(pb.value, dtype=torch.uint8, device=env.device)
    st["counters"] = torch.zeros(nc.value + 4, dtype=torch.int32, device=env.device)
    st["n_counters"] = nc.value
    st["state"] = torch.zeros(4, dtype=torch.int32, device=env.device)
    st["err"] = torch.zeros(4, dtype=torch.int32, device=env.device)
    return st


def ll_applies(st, M_total: int, N: int, K: int, a: torch.Tensor, b: torch.Tensor) -> bool:
    return (st is not None and M_total <= st["rows"] and K >= 128 and K % 64 == 0 and N % 8 == 0
            and a.dtype in _DT and a.dtype == b.dtype and b.stride(1) == 1 and b.stride(0) % 8 == 0
            and b.data_ptr() % 16 == 0)


def check_error(st, op: str) -> None:
    if st is not None and int(st["err"][0]) != 0:
        raise RuntimeError(f"{op} (low-latency form): a device-side wait timed out (peer data never arrived)")


def ag_gemm_ll(st, env, rank: int, a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, num_ctas: int) -> torch.Tensor:
    M_per_rank, K = a.shape
    N_local = b.shape[0]
    a = a.contiguous()
    _lib.check(
        _lib.hip().tdist_ag_gemm_ll(ctypes.byref(env.heap.ctx_struct()), _lib.ptr(a), _lib.ptr(b), _lib.ptr(c),
                                    _lib.ptr(st["data"][0][rank]), _lib.ptr(st["data"][1][rank]),
                                    _lib.ptr(st["flags"][0][rank]), _lib.ptr(st["flags"][1][rank]),
                                    _lib.ptr(st["partial"]), ctypes.c_longlong(st["partial"].numel()),
                                    _lib.ptr(st["counters"]), st["n_counters"], _lib.ptr(st["err"]),
                                    _lib.ptr(st["state"]), ctypes.c_longlong(a.stride(0)),
                                    ctypes.c_longlong(b.stride(0)), ctypes.c_longlong(c.stride(0)), M_per_rank,
                                    N_local, K, ctypes.c_longlong(st["rows"]), num_ctas, _DT[a.dtype],
                                    _lib.stream_handle()), "tdist_ag_gemm_ll")
    return c


def gemm_rs_ll(st, env, rank: int, num_ranks: int, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor,
               num_ctas: int) -> torch.Tensor:
    M, K_local = a.shape
    N = b.shape[0]
    a = a.contiguous()
    _lib.check(
        _lib.hip().tdist_gemm_rs_ll(ctypes.byref(env.heap.ctx_struct()), _lib.ptr(a), _lib.ptr(b), _lib.ptr(out),
                                    _lib.ptr(st["data"][0][rank]), _lib.ptr(st["data"][1][rank]),
                                    _lib.ptr(st["flags"][0][rank]), _lib.ptr(st["flags"][1][rank]),
                                    _lib.ptr(st["partial"]), ctypes.c_longlong(st["partial"].numel()),
                                    _lib.ptr(st["counters"]), st["n_counters"], _lib.ptr(st["err"]),
                                    _lib.ptr(st["state"]), ctypes.c_longlong(a.stride(0)),
                                    ctypes.c_longlong(b.stride(0)), ctypes.c_longlong(out.stride(0)),
                                    M // num_ranks, N, K_local, ctypes.c_longlong(st["data"][0][rank].numel()),
                                    num_ctas, _DT[a.dtype], _lib.stream_handle()), "tdist_gemm_rs_ll")
    return out
