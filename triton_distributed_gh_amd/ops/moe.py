"""Tensor-parallel MoE ops: AllGather + grouped GEMM and grouped GEMM + topk-reduce + ReduceScatter,
plus the index utilities around them.

API of the reference:
* ``create_ag_group_gemm_context`` / ``ag_group_gemm(a, b, ctx, full_topk_ids)`` (reference:
  python/triton_dist/kernels/nvidia/allgather_group_gemm.py:337-411)
* ``create_moe_rs_context`` / ``run_moe_reduce_rs(x, weights, chosen_experts, expert_weight, ctx,
  n_chunks)`` (reference: kernels/nvidia/moe_reduce_rs.py:87-119, :569-656)
* ``moe_utils``: ``calc_gather_scatter_index``, ``reduce_topk``, histogram (reference:
  kernels/nvidia/moe_utils.py:95-405) and ``moe_ag_scatter_align_block_size`` (csrc/lib/moe_utils.cu)

Expert weights are consumed K-contiguous: ``[E, N, K]``.  The reference's ``[E, K, N]`` layout is
accepted and converted once per weight tensor (cached).
"""
from __future__ import annotations

import ctypes
import os
import dataclasses
from typing import Any, Optional

import torch

from .. import _lib
from ..utils.distributed import default_num_ctas, forbid_graph_capture, get_env
from ..autotuner import tuning
from .ag_gemm import (AllGatherGEMMTensorParallelContext, EPOCH_MOD, _push_kernel, _rows_per_chunk,
                      create_ag_gemm_intra_node_context)
from .gemm import _DT

BLOCK_M = 256


# ---- index utilities ----------------------------------------------------------------------------
@dataclasses.dataclass(repr=False)
class AlignedIndex:
    sorted_ids: torch.Tensor        # [cap] flat (token*topk+k) or -1
    expert_of_block: torch.Tensor   # [cap/256]
    seg_range: torch.Tensor         # [cap/256, 2]
    order: torch.Tensor             # [cap/256]
    meta: torch.Tensor              # [2+E]: num_blocks, padded rows, tokens per expert


def moe_align_block_size(topk_ids: torch.Tensor, num_experts: int, world_size: int = 1, rank: int = 0,
                         tokens_per_rank: Optional[int] = None) -> AlignedIndex:
    """Sort (token, k) pairs by (expert, source rank) and pad every expert to 256 rows."""
    ids = topk_ids.to(torch.int32).contiguous()
    ntokens, topk = ids.shape
    n = ntokens * topk
    tokens_per_rank = tokens_per_rank or max(1, ntokens // world_size)
    cap = (n + num_experts * (BLOCK_M - 1) + BLOCK_M - 1) // BLOCK_M * BLOCK_M
    dev = ids.device
    if not ids.is_cuda:
        return _align_cpu(ids, num_experts, world_size, rank, tokens_per_rank, cap)
    nb = cap // BLOCK_M
    z = torch.zeros(4 * nb + 2 + num_experts, dtype=torch.int32, device=dev)        # (one fill for the four zeroed arrays)
    out = AlignedIndex(torch.empty(cap, dtype=torch.int32, device=dev), z[:nb], z[nb:3 * nb].view(nb, 2), z[3 * nb:4 * nb],
                       z[4 * nb:])
    lib = _lib.hip()
    part = torch.empty(max(1, lib.tdist_moe_align_part_ints(n, num_experts, world_size)), dtype=torch.int32, device=dev)
    _lib.check(
        lib.tdist_moe_align(_lib.ptr(ids), _lib.ptr(out.sorted_ids), _lib.ptr(out.expert_of_block),
                            _lib.ptr(out.seg_range), _lib.ptr(out.order), _lib.ptr(out.meta), n, num_experts,
                            world_size, topk, tokens_per_rank, rank, cap, _lib.stream_handle(), _lib.ptr(part)),
        "tdist_moe_align")
    return out


def _align_cpu(ids, E, w, rank, tpr, cap):
    """Reference implementation (also the oracle of the GPU kernel's tests)."""
    ntokens, topk = ids.shape
    flat_e = ids.reshape(-1).long()
    src = (torch.arange(ntokens * topk) // topk // tpr).clamp(max=w - 1)
    key = flat_e * w + src
    order = torch.sort(key, stable=True).indices
    counts = torch.bincount(flat_e, minlength=E)
    padded = (counts + BLOCK_M - 1) // BLOCK_M * BLOCK_M
    pad_off = torch.cumsum(padded, 0) - padded
    sorted_ids = torch.full((cap,), -1, dtype=torch.int32)
    start = torch.cumsum(counts, 0) - counts
    for e in range(E):
        c = int(counts[e])
        sorted_ids[int(pad_off[e]): int(pad_off[e]) + c] = order[int(start[e]): int(start[e]) + c].int()
    nb = cap // BLOCK_M
    nblocks = int(padded.sum()) // BLOCK_M
    eob = torch.zeros(nb, dtype=torch.int32)
    seg = torch.zeros((nb, 2), dtype=torch.int32)
    for b in range(nblocks):
        rows = sorted_ids[b * BLOCK_M:(b + 1) * BLOCK_M]
        valid = rows[rows >= 0].long()
        eob[b] = int(flat_e[valid[0]])
        s = src[valid]
        seg[b, 0], seg[b, 1] = int(s.min()), int(s.max())
    stage = [max((s - rank + w) % w for s in range(int(seg[b, 0]), int(seg[b, 1]) + 1)) for b in range(nblocks)]
    ordv = torch.zeros(nb, dtype=torch.int32)
    # eight contiguous ranges of blocks (one per XCD), each walked in arrival order
    perm = []
    for q in range(8):
        lo, hi = q * nblocks // 8, (q + 1) * nblocks // 8
        perm += sorted(range(lo, hi), key=lambda b: (stage[b], b))
    ordv[:nblocks] = torch.tensor(perm, dtype=torch.int32)
    meta = torch.zeros(2 + E, dtype=torch.int32)
    meta[0], meta[1] = nblocks, int(padded.sum())
    meta[2:] = counts.int()
    return AlignedIndex(sorted_ids, eob, seg, ordv, meta)


def moe_align_reference(topk_ids: torch.Tensor, num_experts: int, world_size: int = 1, rank: int = 0,
                         tokens_per_rank: Optional[int] = None) -> AlignedIndex:
    """The native C++ CPU oracle (csrc/runtime/moe_align_ref.cc) of ``moe_align_block_size``."""
    ids = topk_ids.to(torch.int32).cpu().contiguous()
    ntokens, topk = ids.shape
    n = ntokens * topk
    tpr = tokens_per_rank or max(1, ntokens // world_size)
    cap = (n + num_experts * (BLOCK_M - 1) + BLOCK_M - 1) // BLOCK_M * BLOCK_M
    nb = cap // BLOCK_M
    out = AlignedIndex(torch.empty(cap, dtype=torch.int32), torch.zeros(nb, dtype=torch.int32),
                       torch.zeros((nb, 2), dtype=torch.int32), torch.zeros(nb, dtype=torch.int32),
                       torch.zeros(2 + num_experts, dtype=torch.int32))
    rc = _lib.host().tdh_moe_align_reference(_lib.ptr(ids), n, num_experts, world_size, topk, tpr, rank, cap, BLOCK_M,
                                             _lib.ptr(out.sorted_ids), _lib.ptr(out.expert_of_block),
                                             _lib.ptr(out.seg_range), _lib.ptr(out.order), _lib.ptr(out.meta))
    if rc != 0:
        raise ValueError(f"moe_align_reference: invalid input (code {rc})")
    return out


def calc_gather_scatter_index(choosed_experts: torch.Tensor, num_experts: int):
    """(gather_index, scatter_index, splits): rows sorted by expert (stable)."""
    flat = choosed_experts.reshape(-1).long()
    gather_index = torch.sort(flat, stable=True).indices.int()
    scatter_index = torch.empty_like(gather_index)
    scatter_index[gather_index.long()] = torch.arange(flat.numel(), device=flat.device, dtype=torch.int32)
    return gather_index, scatter_index, torch.bincount(flat, minlength=num_experts).int()


def histogram(x: torch.Tensor, bins: int) -> torch.Tensor:
    return torch.bincount(x.reshape(-1).long(), minlength=bins).int()


def reduce_topk(x: torch.Tensor, topk: int, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[ntokens*topk, N] -> [ntokens, N] (optionally weighted), fp32 accumulation."""
    n = x.shape[0] // topk
    if (weights is None and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous()
            and x.shape[1] % 8 == 0 and x.data_ptr() % 16 == 0 and n > 0):
        out = torch.empty((n, x.shape[1]), dtype=x.dtype, device=x.device)      # one launch instead of convert + sum + convert
        _lib.check(_lib.hip().tdist_moe_topk_sum_bf16(_lib.ptr(x), _lib.ptr(out), n, topk, x.shape[1], _lib.stream_handle()),
                   "tdist_moe_topk_sum_bf16")
        return out
    xf = x.float().view(n, topk, -1)
    if weights is not None:
        xf = xf * weights.float().view(n, topk, 1)
    return xf.sum(1).to(x.dtype)


_WEIGHT_CACHE = {}


def _as_enk(b: torch.Tensor, N: int, K: int) -> torch.Tensor:
    """Expert weights as [E, N, K] contiguous."""
    if b.shape[1] == N and b.shape[2] == K and b.is_contiguous():
        return b   # (square weights are taken as [E, N, K])
    key = (b.data_ptr(), b._version, tuple(b.shape))
    if key not in _WEIGHT_CACHE:
        _WEIGHT_CACHE.clear()
        _WEIGHT_CACHE[key] = b.transpose(1, 2).contiguous()
    return _WEIGHT_CACHE[key]


# ---- AllGather + grouped GEMM -----------------------------------------------------------------
@dataclasses.dataclass(repr=False)
class MoEAllGatherGroupGEMMTensorParallelContext:
    ag: AllGatherGEMMTensorParallelContext
    num_experts: int
    topk: int
    N_per_rank: int
    K: int
    rank: int
    num_ranks: int
    env: Any = None
    num_ctas: int = 0
    tickets: Optional[torch.Tensor] = None      # int32 [8]: per-XCD work queues of the grouped GEMM
    last_index: Any = None                      # the AlignedIndex of the latest ag_group_gemm call

    def check_error(self):
        self.ag.check_error()


def create_ag_group_gemm_context(max_ntokens, N_per_rank, K, num_experts, topk, dtype, rank, num_ranks,
                                 num_local_ranks=8, BLOCK_SIZE_M=256, BLOCK_SIZE_N=256, BLOCK_SIZE_K=64,
                                 GROUP_SIZE_M=8, stages=2, num_warps=8, num_ctas=0, ag_streams=None, env=None):
    env = env or get_env()
    num_ctas = default_num_ctas(env, num_ranks, num_ctas)
    ag = create_ag_gemm_intra_node_context(max_ntokens, N_per_rank, K, dtype, dtype, rank, num_ranks,
                                           producer="kernel", num_ctas=num_ctas, ag_streams=ag_streams, env=env)
    # eight work-queue tickets, then the in-kernel producer's per-chunk counters (csrc/ops/moe.hip)
    max_chunks = max(1, (max_ntokens // num_ranks + 15) // 16)
    # (+ the finished-workgroup counter behind them; the kernel leaves all of them zero after every launch)
    tickets = torch.zeros(8 + max_chunks + 4, dtype=torch.int32, device=env.device) if env.backend == "hip" else None
    return MoEAllGatherGroupGEMMTensorParallelContext(ag, num_experts, topk, N_per_rank, K, rank, num_ranks, env, num_ctas,
                                                      tickets)


def ag_group_gemm(a: torch.Tensor, b: torch.Tensor, ctx: MoEAllGatherGroupGEMMTensorParallelContext,
                  full_topk_ids: torch.Tensor) -> torch.Tensor:
    """a [ntokens/w, K] local tokens, b expert weights ([E, N/w, K] or [E, K, N/w]), full_topk_ids
    [ntokens, topk] (already all-gathered).  Returns [ntokens*topk, N/w]: row token*topk+k is
    token's activation through expert full_topk_ids[token, k]."""
    ag, env = ctx.ag, ctx.env
    w, rank = ctx.num_ranks, ctx.rank
    if env.backend != "hip":
        forbid_graph_capture("ag_group_gemm", w)
    M_per_rank, K = a.shape
    ntokens, topk = full_topk_ids.shape
    assert ntokens == M_per_rank * w and K == ctx.K
    N = ctx.N_per_rank
    b = _as_enk(b, N, K)
    if env.backend == "host" or not a.is_cuda:
        return _ag_group_gemm_host(a, b, ctx, full_topk_ids)
    a = a.contiguous()
    idx = moe_align_block_size(full_topk_ids, ctx.num_experts, w, rank, M_per_rank)
    ctx.last_index = idx        # (one rank: the same sort serves run_moe_reduce_rs, see its ``idx`` argument)
    out = torch.empty((ntokens * topk, N), dtype=a.dtype, device=a.device)
    # The call number lives on the device (ag.state, csrc/include/tdist/call_state.hpp): the kernel picks the
    # workspace / flag parity and the flag value itself, so the launch is the same every time and records into
    # a HIP graph; the host count is kept for the names the context exposes (workspace_tensors).
    parity, epoch = 0, 0
    ag.call_count += 1
    rpc = _rows_per_chunk(M_per_rank, ag.M_PER_CHUNK)
    cur = torch.cuda.current_stream()
    push_wgs = 0
    if w > 1:
        # The all-gather rides in the grouped GEMM's launch: its first push_wgs workgroups store this rank's
        # shard into every rank's workspace (the own one too - tiles read every row from there) and then join
        # the consumers, which start on the own-shard blocks at once.  (A separate push kernel in front of the
        # GEMM cost 80 us before the first tile; one on another stream could never start once the GEMM's
        # workgroups, which fill every CU's register file, spin on the peers' flags.)
        A, flags = ag.workspaces[parity][rank], ag.flags[parity][rank]
        push_wgs = tuning.get("ag_moe.PUSH_WGS", 0) or max(8, min(32, 8 * (w - 1)))
    else:
        A, flags = a, None
    cs = env.heap.ctx_struct() if w > 1 else None
    _lib.check(
        _lib.hip().tdist_moe_group_gemm(_lib.ptr(A), _lib.ptr(b), _lib.ptr(out), _lib.ptr(idx.sorted_ids),
                                        _lib.ptr(idx.expert_of_block), _lib.ptr(idx.seg_range) if w > 1 else None,
                                        _lib.ptr(idx.order) if w > 1 else None, _lib.ptr(idx.meta), None,
                                        _lib.ptr(flags), _lib.ptr(ctx.tickets), _lib.ptr(ag.scratch),
                                        ctypes.c_longlong(A.stride(0)), ctypes.c_longlong(b.stride(1)),
                                        ctypes.c_longlong(out.stride(0)), ctypes.c_longlong(b.stride(0)), N, K, topk, 0,
                                        M_per_rank // rpc, epoch, ctx.num_ctas, _DT[a.dtype], _lib.stream_handle(cur),
                                        ctypes.byref(cs) if cs is not None else None, _lib.ptr(a),
                                        ctypes.c_longlong(a.stride(0)), _lib.ptr(A) if w > 1 else None, M_per_rank, rpc,
                                        push_wgs, _lib.ptr(ag.state) if w > 1 else None,
                                        _lib.ptr(ag.workspaces[1][rank]) if w > 1 else None,
                                        _lib.ptr(ag.flags[1][rank]) if w > 1 else None), "tdist_moe_group_gemm")
    return out


def _grouped_matmul_cpu(rows: torch.Tensor, b_enk: torch.Tensor, flat_ids: torch.Tensor) -> torch.Tensor:
    """rows [R, K] with expert id per row -> [R, N] fp32 (masked per-expert matmul)."""
    out = torch.zeros(rows.shape[0], b_enk.shape[1], dtype=torch.float32, device=rows.device)
    for e in range(b_enk.shape[0]):
        m = flat_ids == e
        if bool(m.any()):
            out[m] = rows[m].float() @ b_enk[e].float().t()
    return out


def _ag_group_gemm_host(a, b_enk, ctx, full_topk_ids):
    """Host backend: same all-gather protocol as AG+GEMM, grouped GEMM in PyTorch."""
    from .ag_gemm import _all_gather_host

    ag = ctx.ag
    parity = ag.call_count & 1
    epoch = ag.call_count % EPOCH_MOD + 1
    ag.call_count += 1
    if ctx.num_ranks > 1:
        a_full = torch.cat(_all_gather_host(a.contiguous(), ag, parity, epoch, a.shape[0]))
    else:
        a_full = a
    topk = full_topk_ids.shape[1]
    rows = a_full.repeat_interleave(topk, dim=0)
    return _grouped_matmul_cpu(rows, b_enk, full_topk_ids.reshape(-1).long()).to(a.dtype)


def _moe_reduce_rs_host(x, b_enk, chosen_experts, expert_weight, ctx):
    lib = _lib.host()
    w, rank = ctx.world_size, ctx.rank
    ntokens, topk = chosen_experts.shape
    part = _grouped_matmul_cpu(x, b_enk, chosen_experts.reshape(-1).long()) * expert_weight.reshape(-1, 1).float()
    part = part.view(ntokens, topk, -1).sum(1).to(x.dtype)
    if w == 1:
        return part
    parity = ctx.call_count & 1
    epoch = ctx.call_count % EPOCH_MOD + 1
    ctx.call_count += 1
    tpr = ntokens // w
    N = part.shape[1]
    for i in range(1, w + 1):
        d = (rank + i) % w
        seg = ctx.scatter[parity][d].view(-1)[: w * tpr * N].view(w, tpr, N)
        seg[rank].copy_(part[d * tpr:(d + 1) * tpr])
        lib.tdh_notify32(ctypes.c_void_p(ctx.flags[parity][d].data_ptr() + 4 * rank), epoch, 1)
    rc = lib.tdh_wait32(ctypes.c_void_p(ctx.flags[parity][rank].data_ptr()), w, 0, epoch, ctypes.c_int64(60000))
    if rc != 0:
        raise RuntimeError(f"moe_reduce_rs(host): rank {rank} timed out waiting for partials")
    seg = ctx.scatter[parity][rank].view(-1)[: w * tpr * N].view(w, tpr, N)
    return seg.float().sum(0).to(x.dtype)


# ---- grouped GEMM + topk reduce + ReduceScatter -----------------------------------------------
@dataclasses.dataclass(repr=False)
class MoEReduceRSContext:
    rank: int
    world_size: int
    max_token_num: int
    hidden_dim: int
    num_experts: int
    topk: int
    dtype: torch.dtype
    scatter: list
    flags: list
    scratch: Optional[torch.Tensor]
    env: Any = None
    num_ctas: int = 0
    call_count: int = 0
    n_chunks_max: int = 8
    side_stream: Any = None      # the reduce of chunk c runs here, next to the grouped GEMM of chunk c + 1
    state: Optional[torch.Tensor] = None     # device-side call number of the unchunked form (call_state.hpp)
    numbering: str = ""          # "device" (unchunked, capturable) or "host" (chunked): one per context

    def check_error(self):
        if self.scratch is not None and int(self.scratch[40]) != 0:
            raise RuntimeError("moe_reduce_rs: a device-side wait timed out")


def create_moe_rs_context(rank, world_size, local_world_size, max_token_num, hidden_dim, num_experts, topk,
                          input_dtype, n_chunks_max=8, num_ctas=0, env=None, rs_stream=None) -> MoEReduceRSContext:
    """``rs_stream``: the stream the chunked form (``n_chunks`` > 1) runs its reduces on; created on first
    use when not given.  Ranks that share one GPU should pass streams that do not share a hardware queue with
    another rank's (a reduce spins on its peers: two of them in one queue would wait for each other)."""
    env = env or get_env()
    num_ctas = default_num_ctas(env, world_size, num_ctas)
    heap = env.heap
    scatter, flags = [], []
    for _ in range(2):
        scatter.append(heap.create_tensor_list([max_token_num, hidden_dim], input_dtype, zero=False))
        flags.append(heap.create_tensor_list([n_chunks_max * world_size], torch.int32, arena="sig"))   # [chunk][src]
    # scratch: [40] sticky error word; per chunk c a block of 64 ints at 64 + 64 c: reduce counters [0, 32),
    # grouped-GEMM tickets [32, 64)
    scratch = (torch.zeros(64 + 64 * n_chunks_max, dtype=torch.int32, device=env.device)
               if env.backend == "hip" else None)
    if env.backend == "hip":
        torch.cuda.synchronize()
    if env.cpu_group is not None:
        torch.distributed.barrier(group=env.cpu_group)
    state = torch.zeros(4, dtype=torch.int32, device=env.device) if env.backend == "hip" else None
    if env.backend == "hip":
        torch.cuda.synchronize()
    return MoEReduceRSContext(rank, world_size, max_token_num, hidden_dim, num_experts, topk, input_dtype, scatter,
                              flags, scratch, env, num_ctas, 0, n_chunks_max, rs_stream, state)


def _column_chunks(N: int, n_chunks: int):
    """Cut N columns into at most n_chunks bands of whole 256-column GEMM tiles: [(first column, width)]."""
    tiles = (N + 255) // 256
    n = max(1, min(n_chunks, tiles))
    per = (tiles + n - 1) // n * 256
    return [(c0, min(per, N - c0)) for c0 in range(0, N, per)]


def run_moe_reduce_rs(x: torch.Tensor, weights: torch.Tensor, chosen_experts: torch.Tensor,
                      expert_weight: torch.Tensor, ctx: MoEReduceRSContext, n_chunks: Optional[int] = None, config=None,
                      idx: Optional["AlignedIndex"] = None):
    """x [ntokens*topk, K/w] (row token*topk+k), weights = expert weights ([E, hidden, K/w] or
    [E, K/w, hidden]), chosen_experts [ntokens, topk], expert_weight [ntokens, topk] routing
    weights.  Returns [ntokens/w, hidden] = ReduceScatter_tokens(sum_k w_k * x_k @ W[e_k]).

    ``n_chunks`` > 1 cuts the hidden dimension into bands of whole GEMM tiles and pipelines them as the
    reference does (kernels/nvidia/moe_reduce_rs.py:238-244, :569-656): the topk reduce + scatter of band c runs
    on a second stream next to the grouped GEMM of band c + 1, which leaves ``moe_rs.RS_CUS`` compute units
    to it (a grouped-GEMM workgroup fills the register file of its CU, so the two kernels cannot share one).
    Default: the ``moe_rs.N_CHUNKS`` tuning knob (1 = one GEMM launch, one reduce launch, back to back)."""
    env = ctx.env
    w, rank = ctx.world_size, ctx.rank
    ntokens, topk = chosen_experts.shape
    K = x.shape[1]
    N = ctx.hidden_dim
    assert x.shape[0] == ntokens * topk and ntokens % w == 0 and ntokens <= ctx.max_token_num
    b = _as_enk(weights, N, K)
    if env.backend == "host" or not x.is_cuda:
        return _moe_reduce_rs_host(x, b, chosen_experts, expert_weight, ctx)
    x = x.contiguous()
    if idx is None:       # (``idx``: the sort of the same ``chosen_experts`` by ``moe_align_block_size(ids, E, 1, 0, ntokens)``)
        idx = moe_align_block_size(chosen_experts, ctx.num_experts, 1, 0, ntokens)
    gemm_out = torch.empty((ntokens * topk, N), dtype=x.dtype, device=x.device)
    scale = expert_weight.reshape(-1).to(torch.float32).contiguous()
    lib = _lib.hip()
    nck = int(n_chunks) if n_chunks is not None else int(tuning.get("moe_rs.N_CHUNKS", 1))
    chunks = _column_chunks(N, min(nck, ctx.n_chunks_max)) if w > 1 else [(0, N)]
    esz = x.element_size()
    # with a reduce running next to it the grouped GEMM leaves some compute units free
    gemm_ctas = ctx.num_ctas
    if len(chunks) > 1:
        budget = ctx.num_ctas or torch.cuda.get_device_properties(x.device).multi_processor_count
        gemm_ctas = max(8, budget - min(int(tuning.get("moe_rs.RS_CUS", 32)), budget // 4))

    def group_gemm(c, c0, nc):
        blk = ctx.scratch[64 + 64 * c:]
        _lib.check(
            lib.tdist_moe_group_gemm(_lib.ptr(x), ctypes.c_void_p(b.data_ptr() + c0 * b.stride(1) * esz),
                                     ctypes.c_void_p(gemm_out.data_ptr() + c0 * esz), _lib.ptr(idx.sorted_ids),
                                     _lib.ptr(idx.expert_of_block), None, None, _lib.ptr(idx.meta), _lib.ptr(scale), None,
                                     _lib.ptr(blk[32:]), _lib.ptr(ctx.scratch[40:]), ctypes.c_longlong(x.stride(0)),
                                     ctypes.c_longlong(b.stride(1)), ctypes.c_longlong(gemm_out.stride(0)),
                                     ctypes.c_longlong(b.stride(0)), nc, K, topk, 1, 1, 0, gemm_ctas, _DT[x.dtype],
                                     _lib.stream_handle(), None, None, ctypes.c_longlong(0), None, 0, 0, 0, None, None,
                                     None),
            "tdist_moe_group_gemm")

    if w == 1:
        group_gemm(0, 0, N)
        from .moe import reduce_topk as _rt

        return _rt(gemm_out, topk)
    # The unchunked form numbers its calls on the device (one kernel node per launch, no memset: it records
    # into a HIP graph); the chunked form runs several launches per call on two streams and takes parity and
    # epoch from the host.  The two counts know nothing of each other, so a context serves one form only.
    numbering = "device" if len(chunks) == 1 else "host"
    if ctx.numbering and ctx.numbering != numbering:
        raise RuntimeError("run_moe_reduce_rs: this context was used with n_chunks "
                           f"{'== 1' if ctx.numbering == 'device' else '> 1'} before; the chunked and the unchunked "
                           "form number their calls differently - create one context per form")
    ctx.numbering = numbering
    if numbering == "host":
        forbid_graph_capture("run_moe_reduce_rs (n_chunks > 1)", w)
    parity = ctx.call_count & 1 if numbering == "host" else 0
    epoch = ctx.call_count % EPOCH_MOD + 1
    ctx.call_count += 1
    out = torch.empty((ntokens // w, N), dtype=x.dtype, device=x.device)
    cs = env.heap.ctx_struct()
    rs_ctas = int(os.environ.get("TDIST_MOE_RS_CTAS", "0")) or ctx.num_ctas

    def reduce_rs(c, c0, nc):
        blk = ctx.scratch[64 + 64 * c:]
        dev = numbering == "device"
        _lib.check(
            lib.tdist_moe_topk_reduce_rs(ctypes.byref(cs), ctypes.c_void_p(gemm_out.data_ptr() + c0 * esz),
                                         ctypes.c_void_p(ctx.scatter[parity][rank].data_ptr() + c0 * esz),
                                         _lib.ptr(ctx.flags[parity][rank][c * w:]), _lib.ptr(blk), _lib.ptr(ctx.scratch[40:]),
                                         ctypes.c_void_p(out.data_ptr() + c0 * esz), ntokens, topk, nc,
                                         ctypes.c_longlong(N), epoch, rs_ctas, _lib.stream_handle(),
                                         _lib.ptr(ctx.state) if dev else None,
                                         _lib.ptr(ctx.scatter[1][rank]) if dev else None,
                                         _lib.ptr(ctx.flags[1][rank]) if dev else None),
            "tdist_moe_topk_reduce_rs")

    if len(chunks) == 1:
        group_gemm(0, 0, N)
        reduce_rs(0, 0, N)
        return out
    main = torch.cuda.current_stream()
    if ctx.side_stream is None:
        ctx.side_stream = getattr(env, "side_stream", None) or torch.cuda.Stream(device=x.device)
    side = ctx.side_stream
    for c, (c0, nc) in enumerate(chunks):
        group_gemm(c, c0, nc)
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            reduce_rs(c, c0, nc)
    main.wait_stream(side)
    return out


# ---- decode: a handful of tokens against their top-k experts ------------------------------------------------
MOE_DECODE_MAX_TOKENS = int(os.environ.get("TDIST_MOE_DECODE_MAX_TOKENS", "16"))


# Once tokens share experts, the (token, expert) pairs are grouped by expert into blocks of sixteen rows
# (csrc/ops/gemm.hip: tdist_moe_group16_fwd): every block streams its expert once.  How far that form is used:
#   * on one rank up to MOE_GROUP_MID_FROM_ROWS rows per expert - there the 64-row form (moe_group_mid_fwd) takes over;
#   * where that form is not available (several ranks with a thin slice of the experts, odd shapes) up to
#     MOE_GROUP16_ROWS_PER_EXPERT rows per expert and MOE_GROUP16_MAX_TOKENS tokens; then the grouped GEMM on 256-row
#     tiles (ag_group_gemm + run_moe_reduce_rs).
# Measured on a Qwen3-30B-A3B layer (9.4 MB per expert: blocks of one expert run next to each other and share it in
# the L2): 384 tokens = 24 rows per expert 467 us against 534 us for the grouped GEMM, 512 tokens = 32 rows 554 against
# 546.  An expert too large for the caches (Mixtral-8x7B: 8 experts of 350 MB, top-2) is streamed from memory again
# for every block, so there the limit is 24 rows per expert on average (blocks of 32 rows from 12 on): 64 tokens 648 us
# against 830 us for the grouped GEMM, 96 tokens 717 against 835, 128 tokens 995 against 850.  For such experts the
# grouped form also beats the per-pair kernels from the first token on (1 / 4 / 16 tokens: 142 / 352 / 480 us against
# 158 / 565 / 2178; profiles/r3/moe_group16.md).
MOE_GROUP16_MAX_TOKENS = int(os.environ.get("TDIST_MOE_GROUP16_MAX_TOKENS", "448"))
MOE_GROUP16_ROWS_PER_EXPERT = int(os.environ.get("TDIST_MOE_GROUP16_ROWS_PER_EXPERT", "28"))
MOE_GROUP16_ROWS_PER_BIG_EXPERT = int(os.environ.get("TDIST_MOE_GROUP16_ROWS_PER_BIG_EXPERT", "24"))
MOE_GROUP16_BIG_EXPERT_BYTES = 32 << 20
# rows per block of the grouped form: 0 = sixteen, or thirty-two (every weight fragment feeds two MFMA row blocks) once
# an expert has more than this many rows on average; 16 / 32 force one.  Qwen3-30B-A3B layer, 16 / 32 rows per block:
# 256 tokens 350 / 359 us, 384 tokens 461 / 391; Mixtral-8x7B layer: 32 tokens 496 / 617, 64 tokens 722 / 648, 96 tokens
# 995 / 717 (grouped GEMM 835), 128 tokens 1226 / 995 (grouped GEMM 850).
MOE_GROUP_BLOCK_ROWS = int(os.environ.get("TDIST_MOE_GROUP_BLOCK_ROWS", "0"))
MOE_GROUP32_FROM_ROWS = int(os.environ.get("TDIST_MOE_GROUP32_FROM_ROWS", "18"))
MOE_GROUP32_FROM_ROWS_BIG = int(os.environ.get("TDIST_MOE_GROUP32_FROM_ROWS_BIG", "12"))
# -1: the grouped form also takes MOE_DECODE_MAX_TOKENS tokens or fewer once the pairs outnumber half of the experts
# (tokens then share experts often enough that streaming an expert once per block beats once per pair: Qwen3-30B-A3B,
# 128 experts, top-8: 8 tokens 115 vs 116 us, 12 tokens 161 vs 143, 16 tokens 207 vs 163); 0: never below the limit;
# 1: always
MOE_GROUP16 = int(os.environ.get("TDIST_MOE_GROUP16", "-1"))


def group16_row_limit(E: int, H: int, Ip: int) -> int:
    """Most (token, expert) pairs the expert-grouped decode form takes for this expert shape."""
    big = 3 * H * Ip * 2 > MOE_GROUP16_BIG_EXPERT_BYTES
    return (MOE_GROUP16_ROWS_PER_BIG_EXPERT if big else MOE_GROUP16_ROWS_PER_EXPERT) * E


def _group16_takes(n: int, topk: int, E: int, Ip: int, H: int = 0) -> bool:
    """Whether ``moe_decode_fwd`` runs ``n`` tokens on the expert-grouped kernels."""
    able = (E <= 1024 and Ip >= 64 and n <= MOE_GROUP16_MAX_TOKENS and n * topk <= group16_row_limit(E, H, Ip))
    if not able:
        return False
    if n > MOE_DECODE_MAX_TOKENS:
        return True
    big = 3 * H * Ip * 2 > MOE_GROUP16_BIG_EXPERT_BYTES
    return MOE_GROUP16 == 1 or (MOE_GROUP16 < 0 and (big or n * topk > max(E // 2, topk)))


def moe_decode_supported(x: torch.Tensor, gate_up: torch.Tensor, down: torch.Tensor, topk: int,
                         max_tokens: Optional[int] = None) -> bool:
    """Shapes ``moe_decode_fwd`` takes: bf16 on the GPU, expert weights ``gate_up [E, 2 Ip, H]`` / ``down [E, H, Ip]``
    contiguous, H a multiple of 32, Ip of 32, and at most ``MOE_DECODE_MAX_TOKENS`` tokens (one launch pair over the
    (token, expert) pairs) or at most ``MOE_GROUP16_MAX_TOKENS`` tokens with on average no more than
    ``MOE_GROUP16_ROWS_PER_EXPERT`` rows per expert (pairs grouped by expert in blocks of sixteen;
    ``group16_row_limit``).  ``max_tokens`` lowers the token limit (callers whose buffers are sized for fewer rows)."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 == gate_up.dtype == down.dtype):
        return False
    n, H = x.shape
    E, two_ip, H2 = gate_up.shape
    Ip = two_ip // 2
    if max_tokens is not None and n > max_tokens:
        return False
    few = 0 < n <= MOE_DECODE_MAX_TOKENS
    grouped = (MOE_DECODE_MAX_TOKENS < n <= MOE_GROUP16_MAX_TOKENS and n * topk <= group16_row_limit(E, H, Ip)
               and E <= 1024 and Ip >= 64 and H % 4 == 0)
    return ((few or grouped) and H2 == H and tuple(down.shape) == (E, H, Ip) and H % 32 == 0 and H >= 128
            and Ip % 32 == 0 and gate_up.is_contiguous() and down.is_contiguous() and x.is_contiguous()
            and n * topk <= 65535)


# form of the down-projection kernel: -1 default (narrow), 0 = 16 columns per workgroup on MFMAs, 1 = 8 columns on the
# vector ALUs (csrc/ops/gemm.hip: tdist_moe_decode_down)
MOE_DOWN_FORM = int(os.environ.get("TDIST_MOE_DOWN_NARROW", "-1"))
# form of the gate_up kernel: -1 default (vector ALUs), 0 = decode GEMM on MFMAs, 1 = vector ALUs (tdist_moe_decode_gate_up)
MOE_GATE_UP_FORM = int(os.environ.get("TDIST_MOE_GATE_UP_DOT", "-1"))


def moe_decode_fwd(x: torch.Tensor, topk_ids: torch.Tensor, topk_weights: torch.Tensor, gate_up: torch.Tensor,
                   down: torch.Tensor, down_form: Optional[int] = None, gate_up_form: Optional[int] = None) -> torch.Tensor:
    """``out[t] = sum_k w[t, k] * down[e] @ (silu(gate[e] x[t]) * (up[e] x[t]))`` with ``e = topk_ids[t, k]``.

    The MoE block of a decode step as two weight-streaming launches (csrc/ops/gemm.hip): one decode GEMM with fused
    SiLU per (token, expert) pair, then one kernel whose eight waves stream the ``down`` rows of the token's experts
    and fold the routing weights and the top-k sum into their reduction (``down_form``: see ``MOE_DOWN_FORM``).  Nothing is sorted, padded to tiles or
    gathered: the grouped-GEMM path (``ag_group_gemm`` + ``run_moe_reduce_rs``) costs about 250 us per layer for
    one token of Qwen3-30B-A3B, these two launches stream the same 75 MB of expert weights.
    Reference role: layers/nvidia/tp_moe.py at decode sizes (it has no special case)."""
    n, H = x.shape
    k = topk_ids.shape[1]
    Ip = gate_up.shape[1] // 2
    ids = topk_ids.to(torch.int32).contiguous()
    wts = topk_weights.to(torch.float32).contiguous()
    out = torch.empty((n, H), dtype=x.dtype, device=x.device)
    lib = _lib.hip()
    if down_form is None and gate_up_form is None and _group16_takes(n, k, gate_up.shape[0], Ip, H):
        # several tokens per expert: four launches over the pairs sorted by expert in blocks of sixteen rows, every
        # expert's weights streamed once per block (Qwen3-30B-A3B layer, 32 / 128 tokens: see profiles/r3/moe_group16.md)
        E = gate_up.shape[0]
        big = 3 * H * Ip * 2 > MOE_GROUP16_BIG_EXPERT_BYTES
        bs = MOE_GROUP_BLOCK_ROWS or (32 if n * k > (MOE_GROUP32_FROM_ROWS_BIG if big else MOE_GROUP32_FROM_ROWS) * E else 16)
        nb_max = n * k // bs + E
        hbuf = torch.empty((nb_max * bs, Ip), dtype=x.dtype, device=x.device)
        y = torch.empty((n * k, H), dtype=torch.float32, device=x.device)
        idx = torch.empty(nb_max * (bs + 1) + 2, dtype=torch.int32, device=x.device)
        _lib.check(lib.tdist_moe_group16_fwd(_lib.ptr(x), _lib.ptr(ids), _lib.ptr(wts), _lib.ptr(gate_up), _lib.ptr(down),
                                             _lib.ptr(hbuf), _lib.ptr(y), _lib.ptr(out), _lib.ptr(idx),
                                             _lib.ptr(idx[nb_max * bs:]), _lib.ptr(idx[nb_max * (bs + 1):]), n, k, H, Ip, E,
                                             nb_max, bs, _lib.stream_handle()), "tdist_moe_group16_fwd")
        return out
    hbuf = torch.empty((n * k, Ip), dtype=x.dtype, device=x.device)
    _lib.check(lib.tdist_moe_decode_gate_up(_lib.ptr(x), _lib.ptr(ids), _lib.ptr(gate_up), _lib.ptr(hbuf), n, k, Ip, H,
                                            gate_up.shape[0], MOE_GATE_UP_FORM if gate_up_form is None else gate_up_form,
                                            _lib.stream_handle()), "tdist_moe_decode_gate_up")
    _lib.check(lib.tdist_moe_decode_down(_lib.ptr(hbuf), _lib.ptr(ids), _lib.ptr(wts), _lib.ptr(down), _lib.ptr(out),
                                         n, k, H, Ip, down.shape[0], MOE_DOWN_FORM if down_form is None else down_form,
                                         _lib.stream_handle()), "tdist_moe_decode_down")
    return out


# ---- a few hundred to a few thousand tokens on one rank: rows grouped by expert in blocks of 64 / 128 -------------
# Measured per layer call (scripts/dev/probe_moe_layer_tokens.py, profiles/r3/moe_group16.md):
#   Qwen3-30B-A3B (128 experts of 9.4 MB, top-8), 64-row blocks, 64 columns per workgroup: 335 - 351 us from 64 to 768
#   tokens, 408 at 1024, 500 at 1536, 575 at 2048 - against 350 (256 tokens) .. 438 (448) for the sixteen-row form and
#   505 / 525 / 564 us at 512 / 1024 / 2048 tokens for the grouped GEMM on 256-row tiles;
#   Mixtral-8x7B (8 experts of 352 MB, top-2): 64-row blocks x 128 columns 630 us from 32 to 128 tokens, 128-row blocks
#   695 - 713 us up to 256 tokens, 992 at 512 - against 495 / 550 / 647 us at 32 / 48 / 64 tokens for the sixteen-row
#   form and 837 / 854 / 910 us at 128 / 256 / 512 tokens for the grouped GEMM.
MOE_GROUP_MID = os.environ.get("TDIST_MOE_GROUP_MID", "1") != "0"
# rows per expert (on average) where this form takes over from the sixteen-row form, and where the grouped GEMM on
# 256-row tiles takes over from it (experts up to MOE_GROUP16_BIG_EXPERT_BYTES / larger ones)
MOE_GROUP_MID_FROM_ROWS = int(os.environ.get("TDIST_MOE_GROUP_MID_FROM_ROWS", "15"))
MOE_GROUP_MID_FROM_ROWS_BIG = int(os.environ.get("TDIST_MOE_GROUP_MID_FROM_ROWS_BIG", "12"))
MOE_GROUP_MID_ROWS_PER_EXPERT = int(os.environ.get("TDIST_MOE_GROUP_MID_ROWS_PER_EXPERT", "100"))
MOE_GROUP_MID_ROWS_PER_BIG_EXPERT = int(os.environ.get("TDIST_MOE_GROUP_MID_ROWS_PER_BIG_EXPERT", "80"))


def moe_group_mid_preferred(n: int, topk: int, E: int, H: int, Ip: int) -> bool:
    """Whether ``n`` tokens are past the point where the 64- / 128-row form beats the sixteen-row decode form."""
    big = 3 * H * Ip * 2 > MOE_GROUP16_BIG_EXPERT_BYTES
    return n * topk > (MOE_GROUP_MID_FROM_ROWS_BIG if big else MOE_GROUP_MID_FROM_ROWS) * E
MOE_GROUP_MID_BLOCK = int(os.environ.get("TDIST_MOE_GROUP_MID_BLOCK", "0"))      # 64 / 128 force a block height
MOE_GROUP_MID_WN = int(os.environ.get("TDIST_MOE_GROUP_MID_WN", "0"))            # 1 / 2 force 64 / 128 columns per workgroup


def moe_group_mid_supported(x: torch.Tensor, gate_up: torch.Tensor, down: torch.Tensor, topk: int) -> bool:
    """Shapes ``moe_group_mid_fwd`` takes: bf16 on the GPU, ``gate_up [E, 2 Ip, H]`` / ``down [E, H, Ip]`` contiguous,
    H and Ip multiples of 64 (>= 128), at most 1024 experts, and no more than ``MOE_GROUP_MID_ROWS_PER_EXPERT`` (large
    experts: ``MOE_GROUP_MID_ROWS_PER_BIG_EXPERT``) rows per expert on average."""
    if not (MOE_GROUP_MID and x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 == gate_up.dtype == down.dtype):
        return False
    n, H = x.shape
    E, two_ip, H2 = gate_up.shape
    Ip = two_ip // 2
    big = 3 * H * Ip * 2 > MOE_GROUP16_BIG_EXPERT_BYTES
    limit = (MOE_GROUP_MID_ROWS_PER_BIG_EXPERT if big else MOE_GROUP_MID_ROWS_PER_EXPERT) * E
    return (n > 0 and H2 == H and tuple(down.shape) == (E, H, Ip) and H % 64 == 0 and H >= 128 and Ip % 64 == 0
            and Ip >= 128 and E <= 1024 and gate_up.is_contiguous() and down.is_contiguous() and x.is_contiguous()
            and n * topk <= limit and n * topk // 32 + E <= 65535)


def moe_group_mid_fwd(x: torch.Tensor, topk_ids: torch.Tensor, topk_weights: torch.Tensor, gate_up: torch.Tensor,
                      down: torch.Tensor, block_rows: Optional[int] = None, wn: Optional[int] = None) -> torch.Tensor:
    """The MoE block (same result as ``moe_decode_fwd``) for token counts where an expert has tens to a few hundred
    rows: pairs sorted by expert into blocks of 64 or 128 rows, activations gathered into that order, every block a
    row pass of the mid-M GEMM against its expert (csrc/ops/moe_group_mid.hip), SiLU * up, the same for ``down``, and
    the weighted sum of each token's top-k rows through the inverse map of the sort.  Six launches on static grids.
    Reference role: layers/nvidia/tp_moe.py at world size 1 (ag_group_gemm + moe_reduce_rs without the collectives)."""
    from .elementwise import silu_mul

    n, H = x.shape
    k = topk_ids.shape[1]
    E, two_ip, _ = gate_up.shape
    Ip = two_ip // 2
    pairs = n * k
    big = 3 * H * Ip * 2 > MOE_GROUP16_BIG_EXPERT_BYTES
    if block_rows is None:
        # 32-row blocks (three workgroups per CU: more weight bytes in flight) while an expert has fewer than 28 rows on
        # average - Qwen3-30B-A3B layer at 256 / 384 tokens 293 / 302 us against 335 / 338 with 64-row blocks, level at 512 -
        # and 128-row blocks only for experts the caches cannot share between blocks
        # (Mixtral-8x7B layer: 32-row blocks 600 / 612 us at 64 / 96 tokens against 634 / 639 with 64-row blocks)
        if big:
            block_rows = MOE_GROUP_MID_BLOCK or (32 if pairs <= 24 * E else (64 if pairs <= 32 * E else 128))
        else:
            block_rows = MOE_GROUP_MID_BLOCK or (32 if pairs <= 28 * E else 64)
    ids = topk_ids.to(torch.int32).contiguous()
    wts = topk_weights.to(torch.float32).contiguous()
    nb_max = pairs // block_rows + E
    rows = nb_max * block_rows
    dev = x.device
    idx = torch.empty(rows + nb_max + pairs + 2, dtype=torch.int32, device=dev)
    sorted_ids, eob, pos, meta = idx[:rows], idx[rows:rows + nb_max], idx[rows + nb_max:rows + nb_max + pairs], idx[rows + nb_max + pairs:]
    xs = torch.empty((rows, H), dtype=x.dtype, device=dev)
    gu = torch.empty((rows, two_ip), dtype=x.dtype, device=dev)
    y = torch.empty((rows, H), dtype=x.dtype, device=dev)
    out = torch.empty((n, H), dtype=x.dtype, device=dev)
    lib, st = _lib.hip(), _lib.stream_handle()
    _lib.check(lib.tdist_moe_group_align(_lib.ptr(ids), pairs, E, _lib.ptr(sorted_ids), _lib.ptr(eob), _lib.ptr(pos),
                                         _lib.ptr(meta), nb_max, block_rows, st), "tdist_moe_group_align")
    _lib.check(lib.tdist_moe_gather_rows(_lib.ptr(x), _lib.ptr(sorted_ids), _lib.ptr(meta), _lib.ptr(xs), k, H, block_rows,
                                         nb_max, st), "tdist_moe_gather_rows")
    wn = wn or MOE_GROUP_MID_WN or (2 if big and block_rows > 32 else 1)
    _lib.check(lib.tdist_moe_group_mid_gemm(_lib.ptr(xs), _lib.ptr(gate_up), _lib.ptr(gu), _lib.ptr(eob), _lib.ptr(meta), nb_max,
                                            block_rows, wn, two_ip, H, st), "tdist_moe_group_mid_gemm")
    h = silu_mul(gu)                 # (rows of unused blocks hold whatever was in memory: nothing reads them)
    _lib.check(lib.tdist_moe_group_mid_gemm(_lib.ptr(h), _lib.ptr(down), _lib.ptr(y), _lib.ptr(eob), _lib.ptr(meta), nb_max,
                                            block_rows, wn, H, Ip, st), "tdist_moe_group_mid_gemm")
    _lib.check(lib.tdist_moe_topk_sum_gather(_lib.ptr(y), _lib.ptr(pos), _lib.ptr(wts), _lib.ptr(out), n, k, H, st),
               "tdist_moe_topk_sum_gather")
    return out


def moe_route_supported(logits: torch.Tensor, topk: int) -> bool:
    return (logits.is_cuda and logits.dtype == torch.bfloat16 and logits.dim() == 2 and logits.stride(1) == 1
            and logits.shape[1] <= 1024 and 0 < topk <= min(64, logits.shape[1]) and logits.shape[0] > 0)


def moe_route(logits: torch.Tensor, topk: int, renormalize: bool = True):
    """Router epilogue in one launch: softmax over the experts (fp32), top-k, optional renormalisation.
    ``logits`` bf16 [tokens, E] -> (weights fp32 [tokens, k], ids int32 [tokens, k]), descending; ties go to the
    lower expert index, the set ``torch.topk`` returns.  Reference: the softmax / topk / normalise lines of
    layers/nvidia/tp_moe.py (five library launches here, ~23 us per MoE layer at decode sizes)."""
    n = logits.shape[0]
    ids = torch.empty((n, topk), dtype=torch.int32, device=logits.device)
    wts = torch.empty((n, topk), dtype=torch.float32, device=logits.device)
    _lib.check(_lib.hip().tdist_moe_route(_lib.ptr(logits), ctypes.c_longlong(logits.stride(0)), n, logits.shape[1], topk,
                                          1 if renormalize else 0, _lib.ptr(ids), _lib.ptr(wts), _lib.stream_handle()),
               "tdist_moe_route")
    return wts, ids
