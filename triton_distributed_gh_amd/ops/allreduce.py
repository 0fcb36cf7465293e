"""AllReduce family over the xGMI full mesh.

API of the reference (reference: python/triton_dist/kernels/allreduce.py:28-61 method enum,
kernels/nvidia/allreduce.py:108-134 ``create_allreduce_ctx``, :1101-1126
``get_auto_allreduce_method``, :1129-1208 ``all_reduce``).  Method mapping on MI355X - there is no
NVLS multicast, no TMA and a tree buys nothing on a one-hop full mesh:

    OneShot, OneShot_TMA, OneShot_Multimem, OneShot_LL, OneShot_Multimem_LL  -> one-shot push kernel
    TwoShot, TwoShot_Multimem, TwoShot_Multimem_ST, DoubleTree               -> two-shot kernel

Inputs larger than the workspace are processed in chunks (not allowed under graph capture, as in
the reference :1179-1208).
"""
from __future__ import annotations

import ctypes
import dataclasses
from enum import IntEnum
from typing import Any, List, Optional

import torch

from .. import _lib
from ..utils.distributed import forbid_graph_capture, get_env

EPOCH_MOD = 1 << 30
_DT = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}
MAX_CTAS = 64


class AllReduceMethod(IntEnum):
    Unknown = 0
    OneShot = 1
    TwoShot = 2
    DoubleTree = 3
    OneShot_TMA = 4
    OneShot_Multimem = 5
    TwoShot_Multimem = 6
    TwoShot_Multimem_ST = 7
    OneShot_LL = 8
    OneShot_Multimem_LL = 9
    AllReduceEnumMax = 10


_ALLREDUCE_METHODS = {
    "double_tree": AllReduceMethod.DoubleTree,
    "one_shot": AllReduceMethod.OneShot,
    "two_shot": AllReduceMethod.TwoShot,
    "one_shot_tma": AllReduceMethod.OneShot_TMA,
    "one_shot_multimem": AllReduceMethod.OneShot_Multimem,
    "two_shot_multimem": AllReduceMethod.TwoShot_Multimem,
    "two_shot_multimem_st": AllReduceMethod.TwoShot_Multimem_ST,
}
_ONE_SHOT = {AllReduceMethod.OneShot, AllReduceMethod.OneShot_TMA, AllReduceMethod.OneShot_Multimem,
             AllReduceMethod.OneShot_LL, AllReduceMethod.OneShot_Multimem_LL}


def to_allreduce_method(method: str) -> AllReduceMethod:
    if method not in _ALLREDUCE_METHODS:
        raise ValueError(f"Invalid method name {method}. Supported methods: {list(_ALLREDUCE_METHODS.keys())}")
    return _ALLREDUCE_METHODS[method]


def get_allreduce_methods():
    return list(_ALLREDUCE_METHODS.keys())


def kernel_kind(method: AllReduceMethod) -> int:
    """1 = one-shot kernel, 2 = two-shot kernel."""
    return 1 if method in _ONE_SHOT else 2


def workspace_bytes_per_in_byte(world_size: int, method: AllReduceMethod) -> int:
    return world_size if kernel_kind(method) == 1 else 2


def get_max_chunk_nbytes(workspace_nbytes: int, world_size: int, method: AllReduceMethod) -> int:
    return workspace_nbytes // workspace_bytes_per_in_byte(world_size, method) // 512 * 512


def get_auto_allreduce_method(nbytes: int) -> AllReduceMethod:
    """Per link, one-shot moves n bytes and two-shot 2n/w; two-shot pays a second flag round trip."""
    return AllReduceMethod.OneShot if nbytes <= 256 * 1024 else AllReduceMethod.TwoShot


@dataclasses.dataclass(repr=False)
class AllReduceContext:
    workspace_nbytes: int
    rank: int
    world_size: int
    local_world_size: int
    workspaces: List[List[torch.Tensor]]     # [parity][rank] uint8 [workspace_nbytes]
    flags: List[List[torch.Tensor]]          # [parity][rank] int32 [2 * w * MAX_CTAS]
    err: Optional[torch.Tensor]
    env: Any = None
    call_count: int = 0                      # host path: selects parity / epoch; HIP path: statistics only
    state: Optional[torch.Tensor] = None     # HIP path: [call number, finished workgroups] int32, owned by the kernels

    def check_error(self) -> None:
        if self.err is not None and int(self.err[0]) != 0:
            raise RuntimeError("all_reduce: a device-side wait timed out")


def create_allreduce_ctx(workspace_nbytes: int, rank: int, world_size: int, local_world_size: Optional[int] = None,
                         env=None) -> AllReduceContext:
    env = env or get_env()
    heap = env.heap
    workspace_nbytes = (workspace_nbytes + 511) // 512 * 512
    ws, fl = [], []
    for _ in range(2):
        ws.append(heap.create_tensor_list([workspace_nbytes], torch.uint8, zero=False))
        fl.append(heap.create_tensor_list([2 * world_size * MAX_CTAS], torch.int32, arena="sig"))
    err = torch.zeros(4, dtype=torch.int32, device=env.device) if env.backend == "hip" else None
    state = torch.zeros(4, dtype=torch.int32, device=env.device) if env.backend == "hip" else None
    if env.backend == "hip":
        torch.cuda.synchronize()
    if env.cpu_group is not None:
        torch.distributed.barrier(group=env.cpu_group)
    return AllReduceContext(workspace_nbytes, rank, world_size, local_world_size or world_size, ws, fl, err, env,
                            state=state)


def _run_straggler(straggler_option, rank):
    if straggler_option and torch.cuda.is_available():
        srank, cycles = straggler_option
        if rank == srank:
            torch.cuda._sleep(int(cycles))


def _all_reduce_chunk(x, out, method, ctx: AllReduceContext, max_sm: int):
    env = ctx.env
    w, rank = ctx.world_size, ctx.rank
    n = x.numel()
    esize = x.element_size()
    if env.backend == "host":
        forbid_graph_capture("all_reduce", w)
    # HIP path: the call number lives in ctx.state on the device (the kernels select parity and
    # flag value themselves), so the launch is the same every time and may be captured in a graph.
    parity = ctx.call_count & 1
    epoch = ctx.call_count % EPOCH_MOD + 1
    ctx.call_count += 1
    kind = kernel_kind(method)
    vec = 16 // esize
    if kind == 1:
        cap = ctx.workspace_nbytes // w // esize // vec * vec
        assert n <= cap, "chunk exceeds one-shot workspace"
    else:
        cap = ((n + w - 1) // w + vec - 1) // vec * vec      # chunk stride
        assert 2 * w * cap * esize <= ctx.workspace_nbytes, "chunk exceeds two-shot workspace"
    if env.backend == "host":
        return _all_reduce_host(x, out, ctx, parity, epoch)
    nbytes = n * esize
    G = max(1, min(MAX_CTAS, (nbytes // (w if kind == 2 else 1)) // 8192))
    if max_sm and max_sm > 0:
        G = min(G, max_sm)
    cs = env.heap.ctx_struct()
    _lib.check(
        _lib.hip().tdist_allreduce(ctypes.byref(cs), _lib.ptr(x), _lib.ptr(out), _lib.ptr(ctx.workspaces[0][rank]),
                                   _lib.ptr(ctx.workspaces[1][rank]), _lib.ptr(ctx.flags[0][rank]),
                                   _lib.ptr(ctx.flags[1][rank]), _lib.ptr(ctx.state), _lib.ptr(ctx.err),
                                   ctypes.c_longlong(n), ctypes.c_longlong(cap), G, _DT[x.dtype], kind,
                                   _lib.stream_handle()),
        "tdist_allreduce")
    return out


def _all_reduce_host(x, out, ctx, parity, epoch):
    lib = _lib.host()
    w, rank = ctx.world_size, ctx.rank
    n, esize = x.numel(), x.element_size()
    flat = x.contiguous().view(-1)
    for i in range(1, w):
        peer = (rank + i) % w
        slot = ctx.workspaces[parity][peer][rank * n * esize:(rank + 1) * n * esize].view(x.dtype)
        slot.copy_(flat)
        lib.tdh_notify32(ctypes.c_void_p(ctx.flags[parity][peer].data_ptr() + 4 * rank), epoch, 1)
    acc = torch.zeros(n, dtype=torch.float32)
    for s in range(w):
        if s == rank:
            acc += flat.float()
            continue
        if lib.tdh_wait32(ctypes.c_void_p(ctx.flags[parity][rank].data_ptr() + 4 * s), 1, 0, epoch,
                          ctypes.c_int64(60000)) != 0:
            raise RuntimeError("all_reduce(host) timed out")
        acc += ctx.workspaces[parity][rank][s * n * esize:(s + 1) * n * esize].view(x.dtype).float()
    out.view(-1).copy_(acc.to(out.dtype))
    return out


def all_reduce(x: torch.Tensor, method: AllReduceMethod = AllReduceMethod.Unknown, ctx: AllReduceContext = None,
               output: Optional[torch.Tensor] = None, max_sm: int = -1, straggler_option=None) -> torch.Tensor:
    assert ctx is not None, "all_reduce needs a context from create_allreduce_ctx"
    assert x.is_contiguous() and x.dtype in _DT
    if output is None:
        output = torch.empty_like(x)
    if ctx.world_size == 1:
        output.copy_(x)
        return output
    if method == AllReduceMethod.Unknown:
        method = get_auto_allreduce_method(x.numel() * x.element_size())
    _run_straggler(straggler_option, ctx.rank)
    esize = x.element_size()
    if ctx.env.backend == "host":
        max_chunk = ctx.workspace_nbytes // ctx.world_size // 16 * 16
    else:
        max_chunk = get_max_chunk_nbytes(ctx.workspace_nbytes, ctx.world_size, method)
    max_elems = max_chunk // esize
    assert max_elems > 0, "workspace too small"
    n = x.numel()
    if n <= max_elems:
        _all_reduce_chunk(x.view(-1), output.view(-1), method, ctx, max_sm)
        return output
    xf, of = x.view(-1), output.view(-1)
    for s in range(0, n, max_elems):
        e = min(n, s + max_elems)
        _all_reduce_chunk(xf[s:e], of[s:e], method, ctx, max_sm)
    return output
