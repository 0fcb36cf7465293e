"""GEMM + ReduceScatter (tensor-parallel row-parallel linear producing sequence-sharded output).

    out[M/w, N] = ReduceScatter_M( a[M, K_local] @ b[N, K_local]^T )

API-compatible with the reference's AMD op
(reference: python/triton_dist/kernels/amd/gemm_reduce_scatter.py:368-446 -
``create_gemm_rs_intra_node_context`` / ``gemm_rs_intra_node``).  The MI355X kernel
(csrc/ops/gemm_rs.hip) is ONE persistent launch that interleaves GEMM tiles, peer stores over xGMI,
per-chunk epoch flags and the fp32 reduction; see that file for the protocol.  The scatter buffers
and flags are double-buffered by call parity: rank A can be at most one call ahead of rank B
(A's call n+1 reduction needs B's call n+1 partials), so call n+2 never overwrites data of call n
that B is still reducing.
"""
from __future__ import annotations

import ctypes
import dataclasses
import os
from typing import Any, List, Optional

import torch

from .. import _lib
from ..autotuner import tuning
from ..utils.distributed import default_num_ctas, forbid_graph_capture, get_env
from . import tp_ll
from .gemm import _DT_EX, gemm_nt

EPOCH_MOD = 4093
DEBUG_RACES = os.environ.get("TDIST_DEBUG_RACES", "0") == "1"      # csrc/include/tdist/race_debug.hpp


@dataclasses.dataclass(repr=False)
class GEMMReduceScatterTensorParallelContext:
    rank: int
    num_ranks: int
    max_M: int
    N: int
    output_dtype: torch.dtype
    scatter: List[List[torch.Tensor]]          # [parity][rank] -> [max_M, N] viewed as [w, M_per_rank, N]
    flags: List[List[torch.Tensor]]            # [parity][rank] -> int32 [w * max chunks]
    scratch: Optional[torch.Tensor]            # int32 counters + ticket, error word at the end
    env: Any = None
    fuse_scatter: bool = True
    transpose_weight: bool = False
    M_PER_CHUNK: int = 256
    num_ctas: int = 0
    call_count: int = 0
    backend: str = "hip"
    state: Optional[torch.Tensor] = None       # HIP path: device-side [call number, finished workgroups]
    ll: Any = None                             # low-latency form for decode-sized calls (ops/tp_ll.py)

    @property
    def scatter_bufs(self):
        return self.scatter[self.call_count & 1]

    @property
    def sync_bufs(self):
        return self.flags[self.call_count & 1]

    def check_error(self) -> None:
        tp_ll.check_error(self.ll, "gemm_rs")
        code = int(self.scratch[-4]) if self.scratch is not None else 0
        if code & 4:
            raise RuntimeError("gemm_rs: a flag held a value that belongs to no call that may touch this parity "
                               "(TDIST_DEBUG_RACES entry check)")
        if code:
            raise RuntimeError("gemm_rs: a device-side wait timed out (peer partials never arrived)")


def create_gemm_rs_intra_node_context(max_M, N, output_dtype, rank, num_ranks, tp_group=None, fuse_scatter=True,
                                      transpose_weight=False, M_PER_CHUNK=256, num_ctas=0, env=None):
    """Collective.  ``max_M`` is the full (pre-scatter) row count, ``N`` the output width."""
    if not fuse_scatter:
        raise NotImplementedError("only the fused-scatter epilogue exists")
    # transpose_weight: the caller's weights are [K_local, N]; gemm_rs_intra_node makes (and keeps) the K-contiguous
    # form the tile engine streams (ops/ag_gemm.py: k_contiguous_weight)
    env = env or get_env()
    assert max_M % num_ranks == 0
    num_ctas = default_num_ctas(env, num_ranks, num_ctas)
    heap = env.heap
    max_chunks = num_ranks * max(1, (max_M // num_ranks + 15) // 16)
    scatter, flags = [], []
    for _ in range(2):
        scatter.append(heap.create_tensor_list([max_M, N], output_dtype, zero=False))
        flags.append(heap.create_tensor_list([max_chunks], torch.int32, arena="sig"))
    scratch = state = None
    if env.backend == "hip":
        scratch = torch.zeros(max_chunks + 16, dtype=torch.int32, device=env.device)
        # [call number, finished workgroups]: owned by the kernels, which pick buffer parity and flag
        # value from it (csrc/include/tdist/call_state.hpp) - launches are replay-safe
        state = torch.zeros(4, dtype=torch.int32, device=env.device)
        torch.cuda.synchronize()
    ll = tp_ll.create_ll_state(env, num_ranks, max_M, N, 0, output_dtype, "rs")
    if env.backend == "hip":
        torch.cuda.synchronize()
    if env.cpu_group is not None:
        torch.distributed.barrier(group=env.cpu_group)
    return GEMMReduceScatterTensorParallelContext(rank=rank, num_ranks=num_ranks, max_M=max_M, N=N,
                                                  output_dtype=output_dtype, scatter=scatter, flags=flags,
                                                  scratch=scratch, env=env, fuse_scatter=fuse_scatter,
                                                  transpose_weight=transpose_weight, M_PER_CHUNK=M_PER_CHUNK,
                                                  num_ctas=num_ctas, backend=env.backend, state=state, ll=ll)


def _rows_per_chunk(M_per_rank: int, want: int) -> int:
    # chunks must be whole 256-row tiles (or the whole shard) so tiles never straddle chunks
    if want > 0 and want % 256 == 0 and M_per_rank % want == 0:
        return want
    return M_per_rank


def gemm_rs_intra_node_op(a, b, ctx: GEMMReduceScatterTensorParallelContext, out=None, scale_a=None, scale_b=None):
    env = ctx.env
    w, rank = ctx.num_ranks, ctx.rank
    M, K_local = a.shape
    N = b.shape[0]
    assert b.shape[1] == K_local and N == ctx.N and M <= ctx.max_M and M % w == 0
    eight = a.element_size() == 1
    assert a.dtype == b.dtype and (ctx.output_dtype == (torch.bfloat16 if eight else a.dtype)), \
        "partials travel in the output dtype (bf16 for 8-bit operands)"
    M_per_rank = M // w
    if out is None:
        out = torch.empty((M_per_rank, N), dtype=ctx.output_dtype, device=a.device)
    if w == 1 and os.environ.get("TDIST_FORCE_FUSED", "0") != "1":
        if eight or scale_a is not None or scale_b is not None:
            from .gemm import gemm_nt_tiles
            return gemm_nt_tiles(a, b, out=out, scale_a=scale_a, scale_b=scale_b)
        gemm_nt(a, b, out=out)
        return out
    if not eight and tp_ll.ll_applies(ctx.ll, M, N, K_local, a, b) and not DEBUG_RACES and out.stride(1) == 1 \
            and out.stride(0) % 8 == 0 and out.data_ptr() % 16 == 0:
        # decode-sized call: mid-M tile body, device-side call numbering, capturable (ops/tp_ll.py)
        return tp_ll.gemm_rs_ll(ctx.ll, env, rank, w, a, b, out, ctx.num_ctas)
    # The kernel numbers its calls on the device and re-arms its own counters, so a launch is ONE kernel
    # node and records into a HIP graph at any shape.  (It used to be preceded by a hipMemsetAsync of the
    # counters: that memset NODE - not the kernel - faulted with an illegal address when the graphs of two
    # ranks replayed decode-sized launches side by side; putting only the functionally redundant memset back
    # in front of the unchanged kernel brings the fault back.  No op of this package that can be captured
    # issues a memset any more.)
    if ctx.backend != "hip":
        forbid_graph_capture("gemm_rs", w)
    parity = ctx.call_count & 1
    epoch = ctx.call_count % EPOCH_MOD + 1
    ctx.call_count += 1
    rpc = _rows_per_chunk(M_per_rank, tuning.get("gemm_rs.M_PER_CHUNK", ctx.M_PER_CHUNK))
    if ctx.backend == "host":
        return _gemm_rs_host(a, b, out, ctx, parity, epoch, scale_a, scale_b)
    a = a.contiguous()
    lib = _lib.hip()
    cstruct = env.heap.ctx_struct()
    _lib.check(
        lib.tdist_gemm_rs(ctypes.byref(cstruct), _lib.ptr(a), _lib.ptr(b), _lib.ptr(ctx.scatter[0][rank]),
                          _lib.ptr(ctx.flags[0][rank]), _lib.ptr(ctx.scratch), _lib.ptr(ctx.scratch[-4:]),
                          _lib.ptr(out), ctypes.c_longlong(a.stride(0)), ctypes.c_longlong(b.stride(0)),
                          ctypes.c_longlong(out.stride(0)), M_per_rank, N, K_local, rpc, epoch, ctx.num_ctas,
                          _DT_EX[a.dtype], _lib.stream_handle(), _lib.ptr(getattr(ctx, "debug_trace", None)),
                          _lib.ptr(ctx.state), _lib.ptr(ctx.scatter[1][rank]), _lib.ptr(ctx.flags[1][rank]),
                          _lib.ptr(scale_a) if scale_a is not None else ctypes.c_void_p(0),
                          _lib.ptr(scale_b) if scale_b is not None else ctypes.c_void_p(0)),
        "tdist_gemm_rs")
    if DEBUG_RACES and not torch.cuda.is_current_stream_capturing():
        # all w partials of this call were reduced: poison them (see ops/ag_gemm.py)
        ctx.scatter[parity][rank].view(torch.int16).fill_(0x7FC0)
    return out


def _gemm_rs_host(a, b, out, ctx, parity, epoch, scale_a=None, scale_b=None):
    lib = _lib.host()
    w, rank = ctx.num_ranks, ctx.rank
    M = a.shape[0]
    N = b.shape[0]
    M_per_rank = M // w
    partial = a.float() @ b.float().t()
    if scale_a is not None:
        partial = partial * scale_a[:, None]
    if scale_b is not None:
        partial = partial * scale_b[None, :]
    partial = partial.to(out.dtype)
    for i in range(1, w + 1):
        d = (rank + i) % w
        seg = ctx.scatter[parity][d].view(-1)[: w * M_per_rank * N].view(w, M_per_rank, N)
        seg[rank].copy_(partial[d * M_per_rank:(d + 1) * M_per_rank])
        lib.tdh_notify32(ctypes.c_void_p(ctx.flags[parity][d].data_ptr() + 4 * rank), epoch, 1)
    rc = lib.tdh_wait32(ctypes.c_void_p(ctx.flags[parity][rank].data_ptr()), w, 0, epoch, ctypes.c_int64(60000))
    if rc != 0:
        raise RuntimeError(f"gemm_rs(host): rank {rank} timed out waiting for partials")
    seg = ctx.scatter[parity][rank].view(-1)[: w * M_per_rank * N].view(w, M_per_rank, N)
    out.copy_(seg.float().sum(dim=0).to(out.dtype))
    return out


def gemm_rs_intra_node(a, b, ctx: GEMMReduceScatterTensorParallelContext, scale_a=None, scale_b=None):
    """a: [M, K_local], b: [N, K_local] ([K_local, N] for a context created with ``transpose_weight=True``).
    Returns [M // world_size, N].

    Operand types: bf16 / fp16 (partials and output of that type) and fp8 e4m3 / e5m2 / int8 (bf16 partials and
    output).  ``scale_a`` [M] / ``scale_b`` [N] (fp32) de-quantise every rank's partial sums before they are
    scattered, so the reduction adds comparable numbers."""
    if ctx.transpose_weight:
        from .ag_gemm import k_contiguous_weight
        b = k_contiguous_weight(b)
    return gemm_rs_intra_node_op(a, b, ctx, scale_a=scale_a, scale_b=scale_b)
