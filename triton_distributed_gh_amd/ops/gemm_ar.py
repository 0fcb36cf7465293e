"""GEMM + AllReduce (row-parallel linear whose output every rank needs in full).

    out[M, N] = sum_r a_r[M, K_local] @ b_r[N, K_local]^T

API of the reference (reference: python/triton_dist/kernels/nvidia/gemm_allreduce.py -
``create_gemm_ar_context`` :94-108, ``create_ll_gemm_ar_context`` :111-120,
``gemm_allreduce_op`` :546-578, ``low_latency_gemm_allreduce_op`` :509-543).  Two kernels (csrc/ops/gemm_ar.hip),
chosen by the row count inside the op: up to ``LL_MAX_M`` rows the low-latency form (mid-M tile body, two-shot per
block), above it the 256x256-tile form (per-XCD queues, the reduction of a group of tiles queued behind the GEMM of
the next group).  Both number their calls on the device, re-arm their own counters and hand the result over in a
plain tensor whose address does not depend on the call parity, so both record into a HIP graph as one kernel node.
"""
from __future__ import annotations

import ctypes
import dataclasses
from typing import Any, List, Optional

import torch

from .. import _lib
import os

from ..autotuner import tuning
from ..utils.distributed import default_num_ctas, forbid_graph_capture, get_env
from .gemm import _DT, gemm_nt

EPOCH_MOD = 1 << 30
_TILE = 256
# rows up to which the low-latency form (mid-M tile body, two-shot per block, device-side call numbering) runs
LL_MAX_M = int(os.environ.get("TDIST_GEMM_AR_LL_MAX_M", "256"))
# column width of the tiled form's macro-tile: 0 = chosen per call (128 when there are fewer 256-wide tiles than workgroups)
TILE_BN = int(os.environ.get("TDIST_GEMM_AR_BN", "0"))


@dataclasses.dataclass(repr=False)
class GemmARContext:
    rank: int
    world_size: int
    max_M: int
    N: int
    dtype: torch.dtype
    slots: List[List[torch.Tensor]]        # [parity][rank] partial-tile slots
    outs: List[List[torch.Tensor]]         # [parity][rank] symmetric output [max_M, N]
    flags: List[List[torch.Tensor]]        # [parity][rank]
    scratch: Optional[torch.Tensor]        # [0:12] queue tickets + finished-workgroup counter, [12] error word, [16] call number
    env: Any = None
    ar_stream: Any = None
    NUM_COMM_SMS: int = 0
    num_ctas: int = 0
    call_count: int = 0
    ll: Any = None                         # low-latency form: dict of its buffers (hip back-end only)
    result: Optional[torch.Tensor] = None  # context-owned plain [max_M, N] result buffer (copy_to_local=False)

    # reference attribute names
    @property
    def symm_ar_out_buf(self):
        return self.outs[self.call_count & 1][self.rank]

    def check_error(self) -> None:
        if self.scratch is not None and int(self.scratch[12]) != 0:
            raise RuntimeError("gemm_allreduce: a device-side wait timed out")
        if self.ll is not None and int(self.ll["err"][0]) != 0:
            raise RuntimeError("gemm_allreduce (low-latency form): a device-side wait timed out")


def create_gemm_ar_context(ar_stream, rank, world_size, local_world_size, max_M, N, dtype, MIN_BLOCK_SIZE_M=16,
                           MIN_BLOCK_SIZE_N=16, NUM_COMM_SMS=0, num_ctas=0, env=None) -> GemmARContext:
    env = env or get_env()
    assert local_world_size == world_size, "single-node only"
    num_ctas = default_num_ctas(env, world_size, num_ctas)
    heap = env.heap
    tiles = ((max_M + _TILE - 1) // _TILE) * ((N + _TILE - 1) // _TILE)
    tpo = (tiles + world_size - 1) // world_size
    nflags, slot_elems = tpo * world_size + tiles * 8, tpo * _TILE * _TILE
    if env.backend == "hip":
        # flag1 [tiles per owner][w] + flag2 [tiles][8 slabs] and the slot size, for whichever tiling a call picks
        fi, se = ctypes.c_int(), ctypes.c_longlong()
        _lib.check(_lib.hip().tdist_gemm_ar_sizes(max_M, N, world_size, ctypes.byref(fi), ctypes.byref(se)),
                   "tdist_gemm_ar_sizes")
        nflags, slot_elems = fi.value, se.value
    slots, outs, flags = [], [], []
    for _ in range(2):
        slots.append(heap.create_tensor_list([world_size * slot_elems], dtype, zero=False))
        outs.append(heap.create_tensor_list([max_M, N], dtype, zero=False))
        flags.append(heap.create_tensor_list([max(nflags, 2 * world_size)], torch.int32, arena="sig"))
    scratch = torch.zeros(20, dtype=torch.int32, device=env.device) if env.backend == "hip" else None
    ll = _create_ll_buffers(env, world_size, min(max_M, LL_MAX_M), N, dtype) if env.backend == "hip" else None
    if env.backend == "hip":
        torch.cuda.synchronize()
    if env.cpu_group is not None:
        torch.distributed.barrier(group=env.cpu_group)
    return GemmARContext(rank, world_size, max_M, N, dtype, slots, outs, flags, scratch, env, ar_stream,
                         NUM_COMM_SMS, num_ctas, 0, ll)


def _result_buffer(ctx: GemmARContext, M: int, N: int, fresh: bool, like: torch.Tensor) -> torch.Tensor:
    """Where the kernel leaves the result: a fresh tensor (``copy_to_local`` or stream capture), or the
    context's own buffer - valid until the next call, as the reference's symmetric output is."""
    if fresh:
        return torch.empty((M, N), dtype=like.dtype, device=like.device)
    if ctx.result is None:
        ctx.result = torch.empty((ctx.max_M, ctx.N), dtype=like.dtype, device=like.device)
    return ctx.result.view(-1)[: M * N].view(M, N)


def _create_ll_buffers(env, world_size, max_M, N, dtype):
    """Symmetric slots / outputs / flags (both parities) and the local state of the low-latency form."""
    import ctypes as C

    be, ms, mb, nc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    pb = C.c_longlong()
    _lib.check(_lib.hip().tdist_gemm_ar_ll_sizes(max_M, N, world_size, C.byref(be), C.byref(ms), C.byref(mb),
                                                 C.byref(pb), C.byref(nc)), "tdist_gemm_ar_ll_sizes")
    heap = env.heap
    ll = dict(block_elems=be.value, max_slots=ms.value, max_blocks=mb.value, max_M=max_M, slots=[], outs=[], flags=[])
    for _ in range(2):
        ll["slots"].append(heap.create_tensor_list([world_size * ms.value * be.value], dtype, zero=False))
        ll["outs"].append(heap.create_tensor_list([max_M, N], dtype, zero=False))
        ll["flags"].append(heap.create_tensor_list([ms.value * world_size + mb.value], torch.int32, arena="sig"))
    ll["partial"] = torch.empty(pb.value, dtype=torch.uint8, device=env.device)
    ll["counters"] = torch.zeros(nc.value, dtype=torch.int32, device=env.device)
    ll["state"] = torch.zeros(4, dtype=torch.int32, device=env.device)     # [0] call number, [1] finished workgroups
    ll["err"] = torch.zeros(4, dtype=torch.int32, device=env.device)
    return ll


# A separate low-latency context type is unnecessary here; keep the constructor name.
LLGemmARContext = GemmARContext


def create_ll_gemm_ar_context(rank, world_size, local_world_size, max_M, N, dtype, MIN_BLOCK_SIZE_M=16,
                              MIN_BLOCK_SIZE_N=16, NUM_COMM_SMS=0, num_phases=2, num_ctas=0, env=None):
    return create_gemm_ar_context(None, rank, world_size, local_world_size, max_M, N, dtype, MIN_BLOCK_SIZE_M,
                                  MIN_BLOCK_SIZE_N, NUM_COMM_SMS, num_ctas, env)


def gemm_allreduce_op(ctx: GemmARContext, a: torch.Tensor, b: torch.Tensor, gemm_config=None, copy_to_local=True,
                      USE_MULTIMEM_ST=True) -> torch.Tensor:
    env = ctx.env
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and N == ctx.N and M <= ctx.max_M and a.dtype == b.dtype == ctx.dtype
    w, rank = ctx.world_size, ctx.rank
    if w == 1:
        return gemm_nt(a, b)
    if ctx.ll is not None and M <= ctx.ll["max_M"] and K >= 128 and K % 64 == 0:
        return _gemm_ar_ll(ctx, a, b, copy_to_local)
    if env.backend == "host":
        forbid_graph_capture("gemm_allreduce", w)
        parity = ctx.call_count & 1
        epoch = ctx.call_count % EPOCH_MOD + 1
        ctx.call_count += 1
        return _gemm_ar_host(ctx, a, b, parity, epoch, copy_to_local)
    ctx.call_count += 1
    a = a.contiguous()
    # The result lands in the context's symmetric output, ONE buffer for both call parities (the kernel's drain makes
    # that safe, csrc/ops/gemm_ar.hip), so its address is the same every call - eager, captured or replayed.
    # copy_to_local: the kernel also writes a private copy.
    out_sym = ctx.outs[0][rank]
    local = torch.empty((M, N), dtype=a.dtype, device=a.device) if copy_to_local else None
    cs = env.heap.ctx_struct()
    _lib.check(
        _lib.hip().tdist_gemm_ar(ctypes.byref(cs), _lib.ptr(a), _lib.ptr(b), _lib.ptr(ctx.slots[0][rank]),
                                 _lib.ptr(ctx.slots[1][rank]), _lib.ptr(out_sym),
                                 _lib.ptr(ctx.flags[0][rank]), _lib.ptr(ctx.flags[1][rank]), _lib.ptr(ctx.scratch),
                                 _lib.ptr(ctx.scratch[12:]), _lib.ptr(ctx.scratch[16:]),
                                 _lib.ptr(local) if local is not None else ctypes.c_void_p(0),
                                 ctypes.c_longlong(a.stride(0)), ctypes.c_longlong(b.stride(0)), M, N, K, ctx.num_ctas,
                                 tuning.get("gemm_ar.BN", 0) or TILE_BN, _DT[a.dtype], _lib.stream_handle()), "tdist_gemm_ar")
    return local if local is not None else out_sym.view(-1)[: M * N].view(M, N)


def _gemm_ar_ll(ctx: GemmARContext, a: torch.Tensor, b: torch.Tensor, copy_to_local: bool) -> torch.Tensor:
    """Low-latency form (csrc/ops/gemm_ar.hip, second half).  Calls are numbered on the device, so the launch
    may be captured into a HIP graph; the result is then copied by the kernel into a plain tensor whose address
    does not alternate with the call parity."""
    ll, env, rank = ctx.ll, ctx.env, ctx.rank
    M, K = a.shape
    N = b.shape[0]
    a = a.contiguous()
    # The kernel picks the parity of its symmetric output from the call number it keeps on the device; a host
    # mirror of that number drifts as soon as a launch is captured (counted here, not run) or replayed (run, not
    # counted here).  The result therefore always comes back in a plain tensor the kernel copies it into.
    local = _result_buffer(ctx, M, N, copy_to_local or torch.cuda.is_current_stream_capturing(), a)
    cs = env.heap.ctx_struct()
    _lib.check(
        _lib.hip().tdist_gemm_ar_ll(ctypes.byref(cs), _lib.ptr(a), _lib.ptr(b), _lib.ptr(ll["slots"][0][rank]),
                                    _lib.ptr(ll["slots"][1][rank]), _lib.ptr(ll["outs"][0][rank]),
                                    _lib.ptr(ll["outs"][1][rank]), _lib.ptr(ll["flags"][0][rank]),
                                    _lib.ptr(ll["flags"][1][rank]), ll["max_slots"], ll["max_blocks"],
                                    ctypes.c_longlong(ll["block_elems"]), _lib.ptr(ll["partial"]),
                                    ctypes.c_longlong(ll["partial"].numel()), _lib.ptr(ll["counters"]),
                                    ll["counters"].numel(), _lib.ptr(ll["err"]), _lib.ptr(ll["state"]),
                                    _lib.ptr(local),
                                    ctypes.c_longlong(a.stride(0)), ctypes.c_longlong(b.stride(0)), M, N, K,
                                    ctx.num_ctas, tuning.get("gemm_ar.WN", 0), tuning.get("gemm_ar.SPLITS", 0),
                                    _DT[a.dtype], _lib.stream_handle()), "tdist_gemm_ar_ll")
    return local


def _gemm_ar_host(ctx, a, b, parity, epoch, copy_to_local):
    """CPU emulation: two-shot over flat chunks (rank d owns chunk d)."""
    lib = _lib.host()
    w, rank = ctx.world_size, ctx.rank
    M, N = a.shape[0], b.shape[0]
    partial = (a.float() @ b.float().t()).to(ctx.dtype).reshape(-1)
    n = M * N
    ce = (n + w - 1) // w
    fl = ctx.flags[parity]
    for i in range(w):
        d = (rank + i) % w
        lo, hi = min(d * ce, n), min((d + 1) * ce, n)
        ctx.slots[parity][d][rank * ce: rank * ce + (hi - lo)].copy_(partial[lo:hi])
        lib.tdh_notify32(ctypes.c_void_p(fl[d].data_ptr() + 4 * rank), epoch, 1)
    if lib.tdh_wait32(ctypes.c_void_p(fl[rank].data_ptr()), w, 0, epoch, ctypes.c_int64(60000)) != 0:
        raise RuntimeError("gemm_allreduce(host) timed out in phase 1")
    lo, hi = min(rank * ce, n), min((rank + 1) * ce, n)
    mine = ctx.slots[parity][rank][: w * ce].view(w, ce)[:, : hi - lo].float().sum(0).to(ctx.dtype)
    for i in range(w):
        d = (rank + i) % w
        ctx.outs[parity][d].view(-1)[lo:hi].copy_(mine)
        lib.tdh_notify32(ctypes.c_void_p(fl[d].data_ptr() + 4 * (w + rank)), epoch, 1)
    if lib.tdh_wait32(ctypes.c_void_p(fl[rank].data_ptr() + 4 * w), w, 0, epoch, ctypes.c_int64(60000)) != 0:
        raise RuntimeError("gemm_allreduce(host) timed out in phase 2")
    res = ctx.outs[parity][rank].view(-1)[:n].view(M, N)
    return res.clone() if copy_to_local else res


def low_latency_gemm_allreduce_op(ctx: GemmARContext, a, b, gemm_config=None, copy_to_local=True,
                                  USE_MULTIMEM_ST=True) -> torch.Tensor:
    return gemm_allreduce_op(ctx, a, b, gemm_config, copy_to_local, USE_MULTIMEM_ST)
