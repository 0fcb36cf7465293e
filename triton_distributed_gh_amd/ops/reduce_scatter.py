"""Stand-alone reduce-scatter (and its all-gather twin) over the symmetric heap.

API of the reference's kernels/nvidia/reduce_scatter.py (``ReduceScatter2DContext`` :46-145,
``create_reduce_scater_2d_ctx`` :148-179, ``reduce_scatter_2d_op`` :857-866,
``reduce_scatter_multi_node`` :828-854, ``ring_reduce`` :815-825).  The reference pipelines an
intra-node scatter, a ring reduction and an inter-node exchange; inside one full-mesh xGMI node the
whole operation is one hop (csrc/ops/reduce_scatter.hip): publish the input, flag every peer, read
the own row segment from all ranks at once and add in fp32.
"""
from __future__ import annotations

import ctypes
import dataclasses
from typing import Any, List, Optional

import torch

from .. import _lib
from ..utils.distributed import default_num_ctas, forbid_graph_capture, get_env
from .gemm import _DT

EPOCH_MOD = 4093


@dataclasses.dataclass(repr=False)
class ReduceScatter2DContext:
    max_M: int
    N: int
    rank: int
    world_size: int
    local_world_size: int
    dtype: torch.dtype
    overlap_with_gemm: bool
    scatter: List[List[torch.Tensor]]      # [parity][rank] -> [max_M, N]
    flags: List[List[torch.Tensor]]        # [parity][rank] -> int32 [world]
    err: Optional[torch.Tensor]
    env: Any = None
    num_ctas: int = 0
    call_count: int = 0
    # knobs of the reference context, accepted and ignored (no SM partitioning is needed here)
    num_sync_sms: int = 0
    num_p2p_sms: int = 0
    num_reduction_sms: int = 0
    state: Optional[torch.Tensor] = None           # GPU: {call number, finished workgroups, workgroups done reading}
    dev_flags: Optional[List[torch.Tensor]] = None  # GPU: [rank] -> int32 [4 world] (two halves of published / read flags)

    @property
    def nnodes(self) -> int:
        return self.world_size // self.local_world_size

    @property
    def parity(self) -> int:
        """Input copy of the next call: one buffer on the GPU (the kernel numbers its calls itself and does not end
        before the copy has been read), alternating on the CPU backend."""
        return 0 if self.state is not None else self.call_count & 1

    @property
    def scatter_bufs(self):
        return self.scatter[self.parity]

    def get_scatter_bufs_and_signal_for_each_node(self, input: torch.Tensor, node_id: int = 0):
        M = input.shape[0]
        return self.scatter_bufs[self.rank][:M], self.flags[self.parity][self.rank]

    def check_error(self):
        if self.err is not None and int(self.err[0]) != 0:
            raise RuntimeError("reduce_scatter: a device-side wait timed out")


def create_reduce_scater_2d_ctx(max_M, N, rank, world_size, local_world_size, dtype, overlap_with_gemm=True,
                                num_reduction_sms=15, env=None) -> ReduceScatter2DContext:
    """Collective (spelling of the reference kept)."""
    env = env or get_env()
    assert world_size == local_world_size, "single-node only"
    assert max_M % world_size == 0
    heap = env.heap
    scatter, flags = [], []
    for _ in range(2):
        scatter.append(heap.create_tensor_list([max_M, N], dtype, zero=False))
        flags.append(heap.create_tensor_list([world_size], torch.int32, arena="sig"))
    err = torch.zeros(4, dtype=torch.int32, device=env.device) if env.backend == "hip" else None
    state = dev_flags = None
    if env.backend == "hip":
        state = torch.zeros(4, dtype=torch.int32, device=env.device)
        dev_flags = heap.create_tensor_list([4 * world_size], torch.int32, arena="sig")     # see csrc/ops/reduce_scatter.hip
        torch.cuda.synchronize()
    if env.cpu_group is not None:
        torch.distributed.barrier(group=env.cpu_group)
    return ReduceScatter2DContext(max_M, N, rank, world_size, local_world_size, dtype, overlap_with_gemm, scatter, flags,
                                  err, env, default_num_ctas(env, world_size, 0), num_reduction_sms=num_reduction_sms,
                                  state=state, dev_flags=dev_flags)


create_reduce_scatter_2d_ctx = create_reduce_scater_2d_ctx


def reduce_scatter_2d_op(input: torch.Tensor, ctx: ReduceScatter2DContext, output: Optional[torch.Tensor] = None):
    """input [M, N] (this rank's full partial) -> [M / world, N] summed over ranks (fp32 adds)."""
    env = ctx.env
    w, rank = ctx.world_size, ctx.rank
    M, N = input.shape
    assert M % w == 0 and M <= ctx.max_M and N == ctx.N and input.dtype == ctx.dtype
    Mr = M // w
    if output is None:
        output = torch.empty((Mr, N), dtype=input.dtype, device=input.device)
    if w == 1:
        output.copy_(input)
        return output
    parity = ctx.parity
    epoch = ctx.call_count % EPOCH_MOD + 1
    ctx.call_count += 1
    mine = ctx.scatter[parity][rank]
    stage = mine.view(-1)[: M * N].view(M, N)
    if env.backend == "host":
        forbid_graph_capture("reduce_scatter (host backend)", w)
        if input.data_ptr() != stage.data_ptr():
            stage.copy_(input)
        return _reduce_scatter_host(ctx, parity, epoch, M, N, output)
    # GPU: the call number lives on the device (ctx.state) and the input copy is one buffer, so copy + kernel are the
    # same two nodes every time and record into a HIP graph
    if input.data_ptr() != stage.data_ptr():
        stage.copy_(input)
    assert input.dtype in _DT and (Mr * N) % 8 == 0, "16-bit dtypes, segment a multiple of 8 elements"
    cs = env.heap.ctx_struct()
    _lib.check(
        _lib.hip().tdist_reduce_scatter(ctypes.byref(cs), _lib.ptr(mine), _lib.ptr(ctx.dev_flags[rank]),
                                        _lib.ptr(output), _lib.ptr(ctx.err), ctypes.c_longlong(Mr * N), 0, 0,
                                        _DT[input.dtype], ctx.num_ctas, _lib.stream_handle(), _lib.ptr(ctx.state)),
        "tdist_reduce_scatter")
    return output


def _reduce_scatter_host(ctx, parity, epoch, M, N, output):
    lib = _lib.host()
    w, rank = ctx.world_size, ctx.rank
    Mr = M // w
    for i in range(w):
        lib.tdh_notify32(ctypes.c_void_p(ctx.flags[parity][(rank + i) % w].data_ptr() + 4 * rank), epoch, 1)
    if lib.tdh_wait32(ctypes.c_void_p(ctx.flags[parity][rank].data_ptr()), w, 0, epoch, ctypes.c_int64(60000)) != 0:
        raise RuntimeError(f"reduce_scatter(host): rank {rank} timed out")
    acc = torch.zeros(Mr, N, dtype=torch.float32)
    for s in range(w):
        acc += ctx.scatter[parity][s].view(-1)[: M * N].view(M, N)[rank * Mr:(rank + 1) * Mr].float()
    output.copy_(acc.to(output.dtype))
    return output


def reduce_scatter_multi_node(input, ctx: ReduceScatter2DContext, output=None):
    return reduce_scatter_2d_op(input, ctx, output)


def reduce_scatter_for_each_node(input, ctx: ReduceScatter2DContext, output=None):
    return reduce_scatter_2d_op(input, ctx, output)


def ring_reduce(input: torch.Tensor, output: torch.Tensor, begin_idx: int, num_splits: int, num_sms: int = -1):
    """Local helper of the reference: out = sum of ``num_splits`` row blocks of ``input`` (the
    rotation start ``begin_idx`` does not change the sum)."""
    M = input.shape[0] // num_splits
    acc = input.view(num_splits, M, -1).float().sum(0)
    output.copy_(acc.to(output.dtype))
    return output
