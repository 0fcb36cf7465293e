"""AllGather + GEMM (tensor-parallel column-parallel linear with sequence-sharded input).

    C[M, N_local] = AllGather_M(a[M/w, K]) @ b[N_local, K]^T

API-compatible with the reference's AMD op
(reference: python/triton_dist/kernels/amd/all_gather_gemm.py:316-433 -
``create_ag_gemm_intra_node_context`` / ``ag_gemm_intra_node``), redesigned for an MI355X node:

* producer = full-mesh push over the w-1 xGMI links, by a push kernel (default: device-side
  stores + system-scope flag stores, peers rotated per workgroup so all links are busy) or on the
  SDMA copy engines (one C call issues every copy + flag write);
* consumer = persistent gfx950 MFMA GEMM (csrc/ops/ag_gemm.hip) pulling tiles in arrival order
  and waiting on per-(source, chunk) epoch flags;
* workspaces and flags are double-buffered by call parity, so a call needs no flag reset and no
  cross-rank barrier (the reference does ``fill_(0)`` + ``barrier_all_ipc`` after every GEMM).
  Rank A can be at most one call ahead of rank B: A's push n+2 is ordered after A's GEMM n+1,
  which needs B's push n+1, which B issues after its GEMM n.

On a CPU-only machine the same protocol (workspaces in shared memory, flags as atomics) runs
through the host runtime so the plumbing is testable without a GPU.
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
_BM = 256
DEBUG_RACES = os.environ.get("TDIST_DEBUG_RACES", "0") == "1"      # csrc/include/tdist/race_debug.hpp


@dataclasses.dataclass(repr=False)
class AllGatherGEMMTensorParallelContext:
    rank: int
    num_ranks: int
    max_M: int
    N: int
    K: int
    dtype: torch.dtype
    workspaces: List[List[torch.Tensor]]        # [parity][rank] -> [max_M, K]
    flags: List[List[torch.Tensor]]             # [parity][rank] -> int32 [max chunks]
    epoch_table: Optional[torch.Tensor]
    scratch: Optional[torch.Tensor]             # int32: [0] error word, [4:16] tickets, [16:] push / split-K counters
    env: Any = None
    ag_streams: Optional[List["torch.cuda.Stream"]] = None
    serial: bool = False
    M_PER_CHUNK: int = 256
    BLOCK_M: int = 256
    BLOCK_N: int = 256
    BLOCK_K: int = 64
    stages: int = 2
    autotune: bool = False
    producer: str = "kernel"                    # "kernel" | "copy_engine"
    flag_mode: str = "memcpy"                   # copy-engine flags: "memcpy" | "write_value"
    num_ctas: int = 0                           # 0 = one workgroup per CU
    call_count: int = 0
    backend: str = "hip"
    partial: Optional[torch.Tensor] = None      # split-K fp32 partial tiles (allocated on first use)
    # in-kernel producer only: [call number, finished workgroups] int32, owned by the kernels (the
    # call number selects parity and epoch on the device, so the launch can be replayed from a graph)
    state: Optional[torch.Tensor] = None
    # Buffers a captured HIP graph may still point at: once a call ran under stream capture the
    # context refuses to grow scratch / partial (a replay would write into freed memory); buffers that
    # were outgrown before that are parked here instead of being freed while launches may be queued.
    captured: bool = False
    retired: List[torch.Tensor] = dataclasses.field(default_factory=list)
    ll: Any = None                              # low-latency form for decode-sized calls (ops/tp_ll.py)

    # names the reference exposes on its context
    @property
    def workspace_tensors(self):
        return self.workspaces[self.call_count & 1]

    @property
    def barrier_tensors(self):
        return self.flags[self.call_count & 1]

    def check_error(self) -> None:
        tp_ll.check_error(self.ll, "ag_gemm")
        code = int(self.scratch[0]) if self.scratch is not None else 0
        if code & 4:
            raise RuntimeError("ag_gemm: a flag held a value that belongs to no call that may touch this parity "
                               "(TDIST_DEBUG_RACES entry check)")
        if code:
            raise RuntimeError("ag_gemm: a device-side wait timed out (peer data never arrived)")


def create_ag_gemm_intra_node_context(max_M, N, K, input_dtype, output_dtype, rank, num_ranks, tp_group=None,
                                      ag_streams=None, M_PER_CHUNK=256, BLOCK_M=256, BLOCK_N=256, BLOCK_K=64,
                                      stages=2, serial=False, autotune=False, producer=None, num_ctas=0, env=None):
    """Collective.  ``N`` is the per-rank N of the weight shard; ``max_M`` the gathered row count."""
    env = env or get_env()
    assert max_M % num_ranks == 0, "max_M must be divisible by the number of ranks"
    num_ctas = default_num_ctas(env, num_ranks, num_ctas)
    heap = env.heap
    min_chunk = 16
    max_chunks = num_ranks * max(1, (max_M // num_ranks + min_chunk - 1) // min_chunk)
    workspaces, flags = [], []
    for _ in range(2):
        workspaces.append(heap.create_tensor_list([max_M, K], input_dtype, zero=False))
        flags.append(heap.create_tensor_list([max_chunks], torch.int32, arena="sig"))
    # "kernel": push kernel (device-side stores + system-scope flag), no driver-dependent ordering.
    # "copy_engine": SDMA copies; flags by 4-byte copies (as the reference) or stream write-value.
    producer = producer or os.environ.get("TDIST_AG_PRODUCER", "kernel")
    if producer == "copy_engine" and env.backend == "hip" and env.shared_gpu and env.cpu_group is not None and num_ranks > 1:
        # Several PROCESSES on one GPU: the consumer launch never returns once copies into another process's
        # mapping of the same device are queued (observed with two ranks: both block inside the kernel launch).
        # One process with virtual ranks, and one GPU per process, do not go through that driver path.
        raise RuntimeError("the copy-engine producer is not available when several processes share one GPU; "
                           "use producer='kernel' (default) or one GPU per rank")
    epoch_table = scratch = state = None
    if env.backend == "hip":
        epoch_table = torch.arange(EPOCH_MOD + 2, dtype=torch.int32, device=env.device)
        # sized once for the largest call this context admits (every split-K arrival counter of max_M x N):
        # a later, larger shape must never re-allocate what an earlier captured launch still uses
        scratch = torch.zeros(_scratch_ints(max_M, num_ranks, N, max_chunks), dtype=torch.int32, device=env.device)
        state = torch.zeros(4, dtype=torch.int32, device=env.device)
        if ag_streams is None and num_ranks > 1 and getattr(env, "side_stream", None) is not None:
            ag_streams = [env.side_stream]
        if ag_streams is None and num_ranks > 1:
            # one hardware queue per peer link for the copy engines; the push kernel needs one
            n_streams = num_ranks - 1 if producer == "copy_engine" else 1
            ag_streams = [torch.cuda.Stream(priority=-1) for _ in range(n_streams)]
        torch.cuda.synchronize()
    ll = tp_ll.create_ll_state(env, num_ranks, max_M, N, K, input_dtype, "ag") if producer == "kernel" else None
    if env.backend == "hip":
        torch.cuda.synchronize()
    if env.cpu_group is not None:
        torch.distributed.barrier(group=env.cpu_group)
    return AllGatherGEMMTensorParallelContext(
        rank=rank, num_ranks=num_ranks, max_M=max_M, N=N, K=K, dtype=input_dtype, workspaces=workspaces, flags=flags,
        epoch_table=epoch_table, scratch=scratch, env=env, ag_streams=ag_streams, serial=serial,
        M_PER_CHUNK=M_PER_CHUNK, BLOCK_M=BLOCK_M, BLOCK_N=BLOCK_N, BLOCK_K=BLOCK_K, stages=stages, autotune=autotune,
        producer=producer,
        flag_mode=os.environ.get("TDIST_AG_FLAG_MODE", "write_value" if env.shared_gpu else "memcpy"),
        num_ctas=num_ctas, backend=env.backend, state=state, ll=ll)


def _rows_per_chunk(M_per_rank: int, want: int) -> int:
    if want > 0 and M_per_rank % want == 0:
        return want
    return M_per_rank


def _push_copy_engine(ctx, a, parity, epoch, rpc, include_self):
    heap, lib = ctx.env.heap, _lib.hip()
    w, rank = ctx.num_ranks, ctx.rank
    M_per_rank, K = a.shape
    ws_ptrs = (ctypes.c_uint64 * w)(*[heap.peer_ptr(ctx.workspaces[parity][rank], r) for r in range(w)])
    fl_ptrs = (ctypes.c_uint64 * w)(*[heap.peer_ptr(ctx.flags[parity][rank], r) for r in range(w)])
    streams = (ctypes.c_void_p * len(ctx.ag_streams))(*[s.cuda_stream for s in ctx.ag_streams])
    flag_src = ctx.epoch_table.data_ptr() + 4 * epoch
    _lib.check(
        lib.tdist_ag_push_full_mesh(rank, w, _lib.ptr(a), ws_ptrs, fl_ptrs, ctypes.c_void_p(flag_src),
                                    ctypes.c_longlong(M_per_rank), ctypes.c_longlong(rpc),
                                    ctypes.c_longlong(K * a.element_size()), streams, len(ctx.ag_streams), 1,
                                    0 if ctx.flag_mode == "memcpy" else 1, epoch), "tdist_ag_push_full_mesh")
    if include_self:
        s = ctx.ag_streams[0]
        nbytes = M_per_rank * K * a.element_size()
        dst = ctx.workspaces[parity][rank].data_ptr() + rank * nbytes
        _lib.check(lib.tdist_memcpy_async(ctypes.c_void_p(dst), _lib.ptr(a), ctypes.c_size_t(nbytes), 0,
                                          _lib.stream_handle(s)), "local shard copy")
        cpr = M_per_rank // rpc
        for c in range(cpr):
            fp = ctx.flags[parity][rank].data_ptr() + 4 * (rank * cpr + c)
            _lib.check(lib.tdist_stream_write_value32(_lib.stream_handle(s), ctypes.c_void_p(fp), epoch), "local flag")


def _push_kernel(ctx, a, parity, epoch, rpc, include_self, stream):
    lib = _lib.hip()
    M_per_rank, K = a.shape
    cstruct = ctx.env.heap.ctx_struct()
    chunk_bytes = rpc * K * a.element_size()
    slices = max(1, min(16, chunk_bytes // (256 * 1024)))
    # (its own counters: the consumer kernel's words in ctx.scratch must stay zero between launches)
    if getattr(ctx, "push_counters", None) is None or ctx.push_counters.numel() < M_per_rank // rpc:
        ctx.push_counters = torch.zeros(max(64, M_per_rank // rpc), dtype=torch.int32, device=a.device)
    _lib.check(
        lib.tdist_ag_push_kernel(ctypes.byref(cstruct), _lib.ptr(a), _lib.ptr(ctx.workspaces[parity][ctx.rank]),
                                 _lib.ptr(ctx.flags[parity][ctx.rank]), _lib.ptr(ctx.push_counters),
                                 ctypes.c_longlong(K * a.element_size()), M_per_rank, rpc, slices, epoch,
                                 1 if include_self else 0, _lib.stream_handle(stream)), "tdist_ag_push_kernel")


def ag_gemm_intra_node_op(a, b, c, ctx: AllGatherGEMMTensorParallelContext, scale_a=None, scale_b=None):
    env = ctx.env
    w, rank = ctx.num_ranks, ctx.rank
    M_per_rank, K = a.shape
    N_local = b.shape[0]
    assert K == b.shape[1] == ctx.K, "K mismatch with the context"
    assert M_per_rank * w <= ctx.max_M, "M exceeds the context's max_M"
    assert a.dtype == b.dtype == ctx.dtype
    if w == 1 and os.environ.get("TDIST_FORCE_FUSED", "0") != "1":
        if a.element_size() == 1 or scale_a is not None or scale_b is not None:
            from .gemm import gemm_nt_tiles
            return gemm_nt_tiles(a, b, out=c, scale_a=scale_a, scale_b=scale_b)
        gemm_nt(a, b, out=c)
        return c
    if w > 1 and tp_ll.ll_applies(ctx.ll, M_per_rank * w, N_local, K, a, b) and not DEBUG_RACES:
        # decode-sized call: mid-M tile body, device-side call numbering, capturable (ops/tp_ll.py)
        return tp_ll.ag_gemm_ll(ctx.ll, env, rank, a, b, c, ctx.num_ctas)
    a = a.contiguous()
    # The in-kernel producer numbers its calls on the device (ctx.state); every other producer takes
    # parity and epoch from the host and must not be captured in a graph.
    device_calls = ctx.backend == "hip" and ctx.producer == "kernel" and ctx.state is not None
    # With device-side numbering the launch is one kernel node (no memset node: see ops/gemm_rs.py for what
    # a captured memset did) and records into a HIP graph at any shape, tiles straddling ranks included.
    if not device_calls:
        forbid_graph_capture("ag_gemm", w)
    parity = ctx.call_count & 1
    epoch = ctx.call_count % EPOCH_MOD + 1
    ctx.call_count += 1
    rpc = _rows_per_chunk(M_per_rank, tuning.get("ag_gemm.M_PER_CHUNK", ctx.M_PER_CHUNK))

    if ctx.backend == "host":
        return _ag_gemm_host(a, b, c, ctx, parity, epoch, rpc, scale_a, scale_b)

    lib = _lib.hip()
    cur = torch.cuda.current_stream()
    grid = ctx.num_ctas or _num_cus()
    push_wgs = 0
    if w > 1 and ctx.producer == "kernel":
        # producer work items ride in the consumer launch (see csrc/ops/ag_gemm.hip)
        # one workgroup moves ~12-20 GB/s; 8 per outgoing xGMI link keeps every link saturated
        push_wgs = tuning.get("ag_gemm.PUSH_WGS", 0) or max(8, min(32, 8 * (w - 1)))
    plan = _plan(ctx, M_per_rank, w, N_local, K, a.dtype, grid - push_wgs)
    bm, bn, split_k, tiles = plan.bm, plan.bn, plan.split_k, plan.tiles
    straddle = bool(plan.straddle) and w > 1
    if push_wgs and not tuning.get("ag_gemm.PUSH_WGS", 0):
        # CUs that the consumers' single round leaves idle push as well (up to 64 producers): the shard is out
        # sooner and nobody who could compute is taken away
        idle = grid - push_wgs - tiles * split_k
        if idle > 0:
            push_wgs = min(64, push_wgs + idle)
    used = []
    if w > 1 and ctx.producer not in ("kernel", "none"):
        # ("none": the caller gathers - copy engines, another op - and raises the flags itself)
        for s in ctx.ag_streams:
            s.wait_stream(cur)
        _push_copy_engine(ctx, a, parity, epoch, rpc, straddle)
        used = ctx.ag_streams
        if ctx.serial:
            for s in used:
                cur.wait_stream(s)
            torch.cuda.synchronize()
            if env.cpu_group is not None:
                torch.distributed.barrier(group=env.cpu_group)

    ws = ctx.workspaces[0 if device_calls else parity][rank]
    fl = ctx.flags[0 if device_calls else parity][rank]
    cpr = M_per_rank // rpc
    need = (16 + cpr + (tiles if split_k > 1 else 0) + 3) & ~3
    capturing = torch.cuda.is_current_stream_capturing()
    if ctx.scratch.numel() < need:
        # (only reachable when N_local exceeds the context's N: contexts size scratch for max_M x N)
        _may_grow(ctx, capturing, "scratch")
        torch.cuda.synchronize()                 # the old words are live until queued launches have re-armed them
        ctx.retired.append(ctx.scratch)
        ctx.scratch = torch.zeros(need, dtype=torch.int32, device=env.device)
    partial, partial_bytes = None, 0
    if split_k > 1:
        partial_bytes = plan.partial_bytes
        if ctx.partial is None or ctx.partial.numel() < partial_bytes:
            _may_grow(ctx, capturing, "split-K partial buffer")
            if ctx.partial is not None:
                ctx.retired.append(ctx.partial)
            # one allocation for the worst case of this grid: split-K only runs while tiles * split_k <= grid
            ctx.partial = torch.empty(max(partial_bytes, (ctx.num_ctas or _num_cus()) * _BM * _BM * 4), dtype=torch.uint8,
                                      device=env.device)
        partial = ctx.partial
    ctx.captured = ctx.captured or capturing
    _lib.check(
        lib.tdist_ag_gemm_consumer(ctypes.byref(env.heap.ctx_struct()), _lib.ptr(a), _lib.ptr(ws), _lib.ptr(b),
                                   _lib.ptr(c), _lib.ptr(fl), _lib.ptr(ctx.scratch),
                                   ctypes.c_longlong(ctx.scratch.numel()),
                                   _lib.ptr(partial) if partial is not None else ctypes.c_void_p(0),
                                   ctypes.c_longlong(partial_bytes), ctypes.c_longlong(a.stride(0)),
                                   ctypes.c_longlong(ws.stride(0)), ctypes.c_longlong(b.stride(0)),
                                   ctypes.c_longlong(c.stride(0)), M_per_rank, N_local, K, rpc, epoch,
                                   1 if straddle else 0, push_wgs, split_k, ctx.num_ctas, _DT_EX[a.dtype],
                                   _lib.stream_handle(cur),
                                   _lib.ptr(ctx.state) if device_calls else ctypes.c_void_p(0),
                                   _lib.ptr(ctx.workspaces[1][rank]) if device_calls else ctypes.c_void_p(0),
                                   _lib.ptr(ctx.flags[1][rank]) if device_calls else ctypes.c_void_p(0), bm, bn,
                                   _lib.ptr(scale_a) if scale_a is not None else ctypes.c_void_p(0),
                                   _lib.ptr(scale_b) if scale_b is not None else ctypes.c_void_p(0)),
               "tdist_ag_gemm_consumer")
    # `a` is read by the copy-engine streams; joining them keeps the caller free to reuse it in stream order
    for s in used:
        cur.wait_stream(s)
    if DEBUG_RACES and w > 1 and not capturing:
        # every row of this call has been consumed: poison the parity buffer so that a read of the NEXT call on
        # this parity that is not ordered behind its flag sees NaNs instead of this call's plausible values
        ctx.workspaces[parity][rank].view(torch.int16).fill_(0x7FC0)
    return c


def _scratch_ints(max_M: int, w: int, N_local: int, max_chunks: int) -> int:
    """[0] error word, [1] finished workgroups, [4] push ticket, [8:16] tile-queue tickets, [16:] push counters +
    split-K arrival counters (layout of csrc/ops/ag_gemm.hip)."""
    max_tiles = (-(-max_M // 128) + w) * -(-N_local // 128)       # the smallest macro-tile edge of the tile family
    return (16 + max_chunks + max_tiles + 3) & ~3


def _may_grow(ctx, capturing: bool, what: str) -> None:
    if capturing or ctx.captured:
        raise RuntimeError(f"ag_gemm: this call needs a larger {what} than the context holds, but launches of this "
                           "context were captured into a HIP graph that still points at the old one; create the "
                           "context with the largest shape (max_M, N) or run the largest shape once before capturing")


@dataclasses.dataclass
class _AgPlan:
    bm: int
    bn: int
    split_k: int
    straddle: int
    tiles: int
    partial_bytes: int


def _plan(ctx, M_per_rank: int, w: int, N_local: int, K: int, dtype, consumers: int) -> _AgPlan:
    """Tile shape and K split of the consumer kernel for this call (csrc/ops/ag_gemm.hip: plan_ag): the macro-tile
    of the tile family (256x256, 256x128, 128x256) and the split that fill the consumers' share of the chip in one
    round.  ``ag_gemm.SPLIT_K`` / TDIST_AG_SPLIT_K and ``ag_gemm.TILE`` / TDIST_AG_TILE (e.g. "128x256") override."""
    forced = int(os.environ.get("TDIST_AG_SPLIT_K", "0")) or tuning.get("ag_gemm.SPLIT_K", 0)
    tile = os.environ.get("TDIST_AG_TILE", "") or tuning.get("ag_gemm.TILE", "")
    fbm, fbn = (int(x) for x in tile.split("x")) if tile else (0, 0)
    if ctx is not None and (ctx.BLOCK_M, ctx.BLOCK_N) != (256, 256) and not tile:
        fbm, fbn = ctx.BLOCK_M, ctx.BLOCK_N          # a context created with an explicit tile keeps it
    bm, bn, sk, st, tiles = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    pb = ctypes.c_longlong()
    _lib.check(_lib.hip().tdist_ag_gemm_plan(M_per_rank, w, N_local, K, _DT_EX[dtype], max(consumers, 1), fbm, fbn, forced,
                                             ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sk), ctypes.byref(st),
                                             ctypes.byref(tiles), ctypes.byref(pb)), "tdist_ag_gemm_plan")
    return _AgPlan(bm.value, bn.value, sk.value, st.value, tiles.value, pb.value)


_NUM_CUS = 0


def _num_cus() -> int:
    global _NUM_CUS
    if not _NUM_CUS:
        _NUM_CUS = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    return _NUM_CUS


def _all_gather_host(a, ctx, parity, epoch, rpc):
    """CPU emulation of the push + flag protocol over the shared-memory heap; returns the per-source
    row blocks of the gathered operand (own rows straight from ``a``)."""
    lib = _lib.host()
    w, rank = ctx.num_ranks, ctx.rank
    M_per_rank, K = a.shape
    cpr = M_per_rank // rpc
    for i in range(1, w):
        peer = (rank + i) % w
        for ch in range(cpr):
            r0 = rank * M_per_rank + ch * rpc
            ctx.workspaces[parity][peer][r0:r0 + rpc].copy_(a[ch * rpc:(ch + 1) * rpc])
            lib.tdh_notify32(ctypes.c_void_p(ctx.flags[parity][peer].data_ptr() + 4 * (rank * cpr + ch)), epoch, 1)
    ws = ctx.workspaces[parity][rank]
    blocks = []
    for s in range(w):
        if s == rank:
            blocks.append(a)
            continue
        rc = lib.tdh_wait32(ctypes.c_void_p(ctx.flags[parity][rank].data_ptr() + 4 * s * cpr), cpr, 0, epoch,
                            ctypes.c_int64(60000))
        if rc != 0:
            raise RuntimeError(f"ag_gemm(host): rank {rank} timed out waiting for rank {s}")
        blocks.append(ws[s * M_per_rank:(s + 1) * M_per_rank])
    return blocks


def _ag_gemm_host(a, b, c, ctx, parity, epoch, rpc, scale_a=None, scale_b=None):
    M_per_rank = a.shape[0]
    for s, src in enumerate(_all_gather_host(a, ctx, parity, epoch, rpc)):
        blk = src.float() @ b.float().t()
        if scale_a is not None:
            blk = blk * scale_a[s * M_per_rank:(s + 1) * M_per_rank, None]
        if scale_b is not None:
            blk = blk * scale_b[None, :]
        c[s * M_per_rank:(s + 1) * M_per_rank] = blk.to(c.dtype)
    return c


_TRANSPOSED = {}     # (data_ptr, shape, strides, dtype, version) -> K-contiguous copy of a [K, N] weight


def k_contiguous_weight(b: torch.Tensor) -> torch.Tensor:
    """``b`` is a [K, N] weight (``transe_b=True`` of the reference's ops): the [N, K] K-contiguous form the tile
    engine streams, made once per weight and kept (keyed by storage, layout and version counter, so an in-place
    update of the weight makes a new copy) instead of a 100+ MB transpose in front of every call."""
    if b.stride(0) == 1 and b.dim() == 2:
        return b.t()                           # already K-contiguous underneath: a view
    key = (b.data_ptr(), tuple(b.shape), tuple(b.stride()), b.dtype, b._version)
    hit = _TRANSPOSED.get(key)
    if hit is None:
        if len(_TRANSPOSED) >= 64:
            _TRANSPOSED.clear()
        hit = b.t().contiguous()
        _TRANSPOSED[key] = hit
    return hit


def ag_gemm_intra_node(a, b, transe_b=False, ctx: Optional[AllGatherGEMMTensorParallelContext] = None,
                       scale_a: Optional[torch.Tensor] = None, scale_b: Optional[torch.Tensor] = None):
    """a: [M_per_rank, K], b: [N_per_rank, K] (``transe_b=False``).  Returns [M, N_per_rank].

    Operand types: bf16 / fp16 (output of the same type) and fp8 e4m3 / e5m2 / int8 (bf16 output; the all-gather
    then moves half the bytes per xGMI link).  ``scale_a`` [M] (fp32, one per GATHERED row - every rank passes the
    same full vector) and ``scale_b`` [N_per_rank] de-quantise the output in the epilogue."""
    if ctx is None:
        raise RuntimeError("ag_gemm_intra_node requires a context (symmetric workspaces)")
    if transe_b:
        b = k_contiguous_weight(b)
    M_per_rank = a.shape[0]
    N_local = b.shape[0]
    out_dtype = torch.bfloat16 if a.element_size() == 1 else a.dtype
    c = torch.empty([ctx.num_ranks * M_per_rank, N_local], dtype=out_dtype, device=a.device)
    return ag_gemm_intra_node_op(a, b, c, ctx, scale_a=scale_a, scale_b=scale_b)
