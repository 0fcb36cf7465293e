"""Sequence-parallel attention with all-gathered KV (prefill / long context).

Reference: kernels/nvidia/sp_ag_attention_intra_node.py - ``create_sp_ag_attention_context_intra_node``
(:59-99) and ``fused_sp_ag_attn_intra_node`` (:432-521).  Every rank holds a sequence shard of Q,
K and V (contiguous or zig-zag: rank r owns half-blocks r and 2w-1-r of each sequence); K/V are
gathered into symmetric buffers in global order and a varlen causal flash attention runs over them.

MI355X design: the gather is a push kernel (same stream, ahead of the attention) that raises one arrival flag per
source rank on every peer; the attention kernel waits on a source's flag only when it first
reaches that rank's slab, so attention over early slabs overlaps the rest of the gather (the
reference joins the copy stream before launching attention).  Buffers and flags are double
buffered by call parity with epoch-valued flags, so no barrier or reset is needed between calls.
"""
from __future__ import annotations

import ctypes
import dataclasses
import math
from typing import Any, List, Optional

import torch

from .. import _lib
from ..utils.distributed import get_env
from .flash_attn import attention_reference, flash_attn_varlen_fwd

EPOCH_MOD = 4093


@dataclasses.dataclass(repr=False)
class SPAllGatherAttentionContextIntraNode:
    rank: int
    world_size: int
    ag_k_buffers: List[torch.Tensor]      # [parity] -> symmetric [batch * max_seqlen_k, kv_heads, D]
    ag_v_buffers: List[torch.Tensor]
    flags: List[torch.Tensor]             # [parity] -> int32 [world] (signal arena)
    attn_output_buffer: torch.Tensor
    scratch: Optional[torch.Tensor]       # int32: [0] push counter, [4] error word
    ag_stream: Any
    env: Any
    call_count: int = 0

    @property
    def ag_k_buffer(self):
        return self.ag_k_buffers[self.call_count & 1]

    @property
    def ag_v_buffer(self):
        return self.ag_v_buffers[self.call_count & 1]

    def check_error(self) -> None:
        if self.scratch is not None and int(self.scratch[4]) != 0:
            raise RuntimeError("sp_ag_attention: a device-side wait timed out (peer KV never arrived)")

    def finalize(self) -> None:
        pass


def create_sp_ag_attention_context_intra_node(batch_size, q_head, kv_head, max_seqlen_k, max_q_shard_len, head_dim,
                                              input_dtype, output_dtype, rank, world_size, device=None, ag_stream=None,
                                              env=None):
    """Collective."""
    env = env or get_env()
    heap = env.heap
    ks, vs, fl = [], [], []
    for _ in range(2):
        ks.append(heap.create_tensor([batch_size * max_seqlen_k, kv_head, head_dim], input_dtype, zero=False))
        vs.append(heap.create_tensor([batch_size * max_seqlen_k, kv_head, head_dim], input_dtype, zero=False))
        fl.append(heap.create_tensor([max(world_size, 4)], torch.int32, arena="sig"))
    out = torch.empty(batch_size * max_q_shard_len, q_head, head_dim, dtype=output_dtype, device=env.device)
    scratch = None
    if env.backend == "hip":
        scratch = torch.zeros(8, dtype=torch.int32, device=env.device)
        if ag_stream is None:
            ag_stream = torch.cuda.Stream(priority=-1)
        torch.cuda.synchronize()
    if env.cpu_group is not None:
        torch.distributed.barrier(group=env.cpu_group)
    return SPAllGatherAttentionContextIntraNode(rank=rank, world_size=world_size, ag_k_buffers=ks, ag_v_buffers=vs,
                                                flags=fl, attn_output_buffer=out, scratch=scratch, ag_stream=ag_stream,
                                                env=env)


def shard_positions(cu_seqlens_q, rank: int, world_size: int, zigzag: bool, device=None) -> torch.Tensor:
    """Global in-sequence position of every local query row (the layout the kernels assume)."""
    cu = [int(x) for x in cu_seqlens_q]
    pos = []
    for b in range(len(cu) - 1):
        Ls = cu[b + 1] - cu[b]
        i = torch.arange(Ls)
        if zigzag:
            half = Ls // 2
            pos.append(torch.where(i < half, rank * half + i, (2 * world_size - 1 - rank) * half + (i - half)))
        else:
            pos.append(rank * Ls + i)
    return torch.cat(pos).to(device) if pos else torch.zeros(0, dtype=torch.long, device=device)


def fused_sp_ag_attn_intra_node(ctx: SPAllGatherAttentionContextIntraNode, q_shard, k_shard, v_shard, output,
                                cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, rank, world_size,
                                is_causal: bool = True, enable_zig_zag: bool = True, softmax_scale=None):
    """q_shard [total_q_shard, q_heads, D]; k/v_shard [total_kv_shard, kv_heads, D];
    ``cu_seqlens_q`` = cumulative SHARD lengths, ``cu_seqlens_k`` = cumulative FULL lengths
    (every sequence is split evenly, so full = world_size * shard).  Writes ``output``."""
    w = world_size
    Tq, Hq, D = q_shard.shape
    Hkv = k_shard.shape[1]
    output = output if output is not None else ctx.attn_output_buffer[:Tq]
    if w == 1:
        o = flash_attn_varlen_fwd(q_shard, k_shard, v_shard, cu_seqlens_q.int(), cu_seqlens_k.int(), max_seqlen_q,
                                  out=output, softmax_scale=softmax_scale, causal=is_causal) if q_shard.is_cuda else \
            output.copy_(attention_reference(q_shard, k_shard, v_shard, cu_seqlens_q, cu_seqlens_k, is_causal,
                                             softmax_scale).to(output.dtype))
        return o
    assert q_shard.is_cuda, "the multi-rank path needs the HIP backend"
    cu_q = cu_seqlens_q.to(torch.int32)
    cu_k = cu_seqlens_k.to(torch.int32)
    assert k_shard.is_contiguous() and v_shard.is_contiguous()
    min_slab = max_seqlen_k // (2 * w if enable_zig_zag else w)
    assert min_slab >= 32 or cu_q.numel() == 2, "per-rank KV slabs must cover at least one 32-token block"
    parity = ctx.call_count & 1
    epoch = ctx.call_count % EPOCH_MOD + 1
    ctx.call_count += 1
    kf, vf, fl = ctx.ag_k_buffers[parity], ctx.ag_v_buffers[parity], ctx.flags[parity]
    # The KV push runs on the SAME stream, ahead of the attention kernel: attention workgroups that
    # spin on a peer's arrival flag keep their CU slots, so a push kernel on another stream could be
    # locked out by them on every rank at once.  Own push first, then attention over the local KV
    # while the peers' slabs are still arriving.
    cur = torch.cuda.current_stream()
    side = cur
    cstruct = ctx.env.heap.ctx_struct()
    _lib.check(
        _lib.hip().tdist_sp_kv_push(ctypes.byref(cstruct), _lib.ptr(k_shard), _lib.ptr(v_shard), _lib.ptr(kf),
                                    _lib.ptr(vf), _lib.ptr(fl), _lib.ptr(ctx.scratch), _lib.ptr(cu_q), _lib.ptr(cu_k),
                                    cu_q.numel() - 1, k_shard.shape[0], ctypes.c_longlong(Hkv * D * k_shard.element_size()),
                                    1 if enable_zig_zag else 0, epoch, _lib.stream_handle(side)), "tdist_sp_kv_push")
    flash_attn_varlen_fwd(q_shard, kf, vf, cu_q, cu_k, max_seqlen_q, out=output, softmax_scale=softmax_scale,
                          causal=is_causal, pos_mode=2 if enable_zig_zag else 1, rank=rank, world_size=w, flags=fl,
                          epoch=epoch, err=ctx.scratch[4:])
    return output


def default_scale(head_dim: int) -> float:
    return 1.0 / math.sqrt(head_dim)
