"""GEMM + AllReduce (row-parallel linear whose output every rank needs in full).

    out[M, N] = sum_r a_r[M, K_local] @ b_r[N, K_local]^T

API of the reference (reference: python/triton_dist/kernels/nvidia/gemm_allreduce.py -
``create_gemm_ar_context`` :94-108, ``create_ll_gemm_ar_context`` :111-120,
``gemm_allreduce_op`` :546-578, ``low_latency_gemm_allreduce_op`` :509-543).  On MI355X both
entry points run the same persistent kernel (csrc/ops/gemm_ar.hip): every context is already
double-buffered by call parity with epoch flags, which is exactly what the reference's low-latency
context adds on top of its normal one.
"""
from __future__ import annotations

import ctypes
import dataclasses
from typing import Any, List, Optional

import torch

from .. import _lib
from ..utils.distributed import default_num_ctas, forbid_graph_capture, get_env
from .gemm import _DT, gemm_nt

EPOCH_MOD = 1 << 30
_TILE = 256


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
    scratch: Optional[torch.Tensor]        # [0:4] ticket, [4] error word
    env: Any = None
    ar_stream: Any = None
    NUM_COMM_SMS: int = 0
    num_ctas: int = 0
    call_count: int = 0

    # reference attribute names
    @property
    def symm_ar_out_buf(self):
        return self.outs[self.call_count & 1][self.rank]

    def check_error(self) -> None:
        if self.scratch is not None and int(self.scratch[4]) != 0:
            raise RuntimeError("gemm_allreduce: a device-side wait timed out")


def create_gemm_ar_context(ar_stream, rank, world_size, local_world_size, max_M, N, dtype, MIN_BLOCK_SIZE_M=16,
                           MIN_BLOCK_SIZE_N=16, NUM_COMM_SMS=0, num_ctas=0, env=None) -> GemmARContext:
    env = env or get_env()
    assert local_world_size == world_size, "single-node only"
    num_ctas = default_num_ctas(env, world_size, num_ctas)
    heap = env.heap
    tiles = ((max_M + _TILE - 1) // _TILE) * ((N + _TILE - 1) // _TILE)
    tpo = (tiles + world_size - 1) // world_size
    slots, outs, flags = [], [], []
    for _ in range(2):
        slots.append(heap.create_tensor_list([world_size * tpo * _TILE * _TILE], dtype, zero=False))
        outs.append(heap.create_tensor_list([max_M, N], dtype, zero=False))
        flags.append(heap.create_tensor_list([max(tpo * world_size + tiles, 2 * world_size)], torch.int32, arena="sig"))
    scratch = torch.zeros(8, dtype=torch.int32, device=env.device) if env.backend == "hip" else None
    if env.backend == "hip":
        torch.cuda.synchronize()
    if env.cpu_group is not None:
        torch.distributed.barrier(group=env.cpu_group)
    return GemmARContext(rank, world_size, max_M, N, dtype, slots, outs, flags, scratch, env, ar_stream,
                         NUM_COMM_SMS, num_ctas)


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
    forbid_graph_capture("gemm_allreduce", w)
    parity = ctx.call_count & 1
    epoch = ctx.call_count % EPOCH_MOD + 1
    ctx.call_count += 1
    out_sym = ctx.outs[parity][rank]
    if env.backend == "host":
        return _gemm_ar_host(ctx, a, b, parity, epoch, copy_to_local)
    a = a.contiguous()
    cs = env.heap.ctx_struct()
    _lib.check(
        _lib.hip().tdist_gemm_ar(ctypes.byref(cs), _lib.ptr(a), _lib.ptr(b), _lib.ptr(ctx.slots[parity][rank]),
                                 _lib.ptr(out_sym), _lib.ptr(ctx.flags[parity][rank]), _lib.ptr(ctx.scratch),
                                 _lib.ptr(ctx.scratch[4:]), ctypes.c_longlong(a.stride(0)),
                                 ctypes.c_longlong(b.stride(0)), M, N, K, epoch, ctx.num_ctas, _DT[a.dtype],
                                 _lib.stream_handle()), "tdist_gemm_ar")
    res = out_sym.view(-1)[: M * N].view(M, N)
    return res.clone() if copy_to_local else res


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
