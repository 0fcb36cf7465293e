"""All-gather over the xGMI full mesh: bulk methods and the low-latency ("fast") variants.

API of the reference:
* ``AllGatherMethod`` / ``get_auto_all_gather_method`` (reference: kernels/nvidia/allgather.py:46-71) -
  ring / NUMA-2D variants exist there for PCIe and multi-node machines; a single MI355X node is a
  one-hop full mesh, so every method maps to full-mesh push or pull.
* ``create_fast_allgather_context`` / ``fast_allgather(symm_buffer, ctx, ..., mode)`` (reference:
  kernels/nvidia/low_latency_allgather.py:798-816, :957-987).  ``symm_buffer`` is a symmetric
  ``[world, ...]`` tensor whose slot ``rank`` holds the local shard; after the call all slots are
  filled.  Modes: ``pull``, ``push_2d`` / ``push_3d`` / ``push_numa_2d`` (-> push),
  ``push_2d_ll`` / ``push_3d_ll`` / ``push_numa_2d_ll`` / ``push_2d_ll_multimem`` and the
  ``*_multinode`` names (-> LL protocol: 4 payload bytes + 4 flag bytes per 8-byte store).
"""
from __future__ import annotations

import ctypes
import dataclasses
import enum
from typing import Any, List, Optional

import torch

from .. import _lib
from ..utils.distributed import forbid_graph_capture, get_env


class AllGatherMethod(enum.Enum):
    Auto = 0
    All2All_IntraNode = 1
    All2All_InterNode = 2
    Ring1D_IntraNode = 3
    Ring2D_IntraNode = 4
    Ring2D_InterNode = 5
    # names used on this hardware
    FullMeshPush = 1
    FullMeshPull = 6


def get_auto_all_gather_method(num_ranks: int, num_local_ranks: int) -> AllGatherMethod:
    return AllGatherMethod.FullMeshPush


_PUSH = {"push_2d", "push_3d", "push_numa_2d", "push"}
_LL = {"push_2d_ll", "push_3d_ll", "push_numa_2d_ll", "push_2d_ll_multimem", "push_numa_2d_ll_multinode",
       "push_2d_ll_multinode", "ll"}


@dataclasses.dataclass(repr=False)
class FastAllGatherContext:
    rank: int
    world_size: int
    max_bytes_per_rank: int
    flags: List[torch.Tensor]
    stage: List[torch.Tensor]          # LL staging, uint64 [w * max_words] (two halves by parity)
    scratch: Optional[torch.Tensor]
    env: Any = None
    signal_target: int = 0
    ll_state: Optional[torch.Tensor] = None    # LL mode: {call number, finished workgroups} on the device
    bulk_state: Optional[torch.Tensor] = None  # push / pull modes: the same pair

    def check_error(self) -> None:
        if self.scratch is not None and int(self.scratch[16]) != 0:
            raise RuntimeError("fast_allgather: a device-side wait timed out")


def create_fast_allgather_context(rank: int, node: int = 0, num_ranks: int = 1, num_nodes: int = 1,
                                  max_buffer_size: int = 2 * 32 * 128 * 128, env=None) -> FastAllGatherContext:
    """``max_buffer_size`` = bytes of ONE rank's shard the LL staging can hold."""
    env = env or get_env()
    heap = env.heap
    w = env.world_size
    words = (max_buffer_size + 3) // 4
    flags = heap.create_tensor_list([4 * w], torch.int32, arena="sig")
    stage = heap.create_tensor_list([2 * w * words], torch.int64, zero=True)
    scratch = torch.zeros(32, dtype=torch.int32, device=env.device) if env.backend == "hip" else None
    ll_state = torch.zeros(4, dtype=torch.int32, device=env.device) if env.backend == "hip" else None
    bulk_state = torch.zeros(4, dtype=torch.int32, device=env.device) if env.backend == "hip" else None
    if env.backend == "hip":
        torch.cuda.synchronize()
    if env.cpu_group is not None:
        torch.distributed.barrier(group=env.cpu_group)
    return FastAllGatherContext(env.rank, w, max_buffer_size, flags, stage, scratch, env, 0, ll_state, bulk_state)


def fast_allgather(symm_buffer: torch.Tensor, ctx: FastAllGatherContext, rank: Optional[int] = None,
                   num_ranks: Optional[int] = None, world_size: Optional[int] = None, nnodes: int = 1,
                   mode: str = "pull") -> torch.Tensor:
    """In place: fills every slot of the symmetric ``[world, ...]`` buffer."""
    env = ctx.env
    w = ctx.world_size
    assert symm_buffer.shape[0] == w and symm_buffer.is_contiguous()
    if w == 1:
        return symm_buffer
    bytes_per_rank = symm_buffer[0].numel() * symm_buffer.element_size()
    if env.backend == "hip" and mode in _LL:
        # The LL form numbers its calls on the device (staging half and flag value come from a counter the
        # kernel advances): one kernel node, the same every time - it records into a HIP graph.
        assert bytes_per_rank <= ctx.max_bytes_per_rank, "shard larger than the LL staging buffer"
        words = (ctx.max_bytes_per_rank + 3) // 4
        cs = env.heap.ctx_struct()
        # staging slots are strided by the ACTUAL word count of this call
        _lib.check(_lib.hip().tdist_allgather_ll(ctypes.byref(cs), _lib.ptr(symm_buffer), _lib.ptr(ctx.stage[ctx.rank]),
                                                 _lib.ptr(ctx.scratch[16:]), ctypes.c_longlong(bytes_per_rank),
                                                 ctypes.c_uint(0), _lib.stream_handle(), _lib.ptr(ctx.ll_state),
                                                 ctypes.c_longlong(w * words)), "tdist_allgather_ll")
        return symm_buffer
    if mode != "pull" and mode not in _PUSH and mode not in _LL:
        raise ValueError(f"unknown all-gather mode {mode}")
    if env.backend == "host":          # (one protocol for every mode on the CPU backend)
        forbid_graph_capture("fast_allgather (host backend)", w)
        ctx.signal_target += 1
        return _allgather_host(symm_buffer, ctx, ctx.signal_target)
    # The bulk forms number their calls on the device too (flag half and flag value from a counter the kernel advances,
    # the slice counters re-armed by the last workgroup): one kernel node that records into a HIP graph.
    cs = env.heap.ctx_struct()
    slices = max(1, min(8, bytes_per_rank // (128 * 1024)))
    _lib.check(_lib.hip().tdist_allgather(ctypes.byref(cs), _lib.ptr(symm_buffer), _lib.ptr(ctx.flags[ctx.rank]),
                                          _lib.ptr(ctx.scratch), _lib.ptr(ctx.scratch[16:]),
                                          ctypes.c_longlong(bytes_per_rank), slices, 0, 1 if mode == "pull" else 0,
                                          _lib.stream_handle(), _lib.ptr(ctx.bulk_state), ctypes.c_longlong(2 * w)),
               "tdist_allgather")
    return symm_buffer


def _allgather_host(symm_buffer, ctx, epoch):
    lib = _lib.host()
    env = ctx.env
    w, rank = ctx.world_size, ctx.rank
    peers = env.heap.peer_views(symm_buffer)
    for i in range(1, w):
        p = (rank + i) % w
        peers[p][rank].copy_(symm_buffer[rank])
        lib.tdh_notify32(ctypes.c_void_p(ctx.flags[p].data_ptr() + 4 * rank), epoch, 1)
    for s in range(w):
        if s != rank and lib.tdh_wait32(ctypes.c_void_p(ctx.flags[rank].data_ptr() + 4 * s), 1, 0, epoch,
                                        ctypes.c_int64(60000)) != 0:
            raise RuntimeError("fast_allgather(host) timed out")
    return symm_buffer


def all_gather_full_mesh(local: torch.Tensor, symm_buffer: torch.Tensor, ctx: FastAllGatherContext,
                         method: AllGatherMethod = AllGatherMethod.Auto) -> torch.Tensor:
    """Bulk all-gather entry: copy ``local`` into slot ``rank`` then push / pull the rest."""
    symm_buffer[ctx.rank].copy_(local)
    return fast_allgather(symm_buffer, ctx, mode="pull" if method == AllGatherMethod.FullMeshPull else "push_2d")
