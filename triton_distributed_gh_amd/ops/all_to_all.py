"""DeepSeek-style low-latency all-to-all (expert-parallel dispatch / combine for decode) and the
fp8 group quantisation that goes with it.

API of the reference (reference: python/triton_dist/kernels/nvidia/low_latency_all_to_all.py -
``create_all_to_all_context`` :176-195, ``fast_all_to_all`` :198-257, ``all_to_all_post_process``
:260-279; fp8 helpers tutorials/04-deepseek-infer-all2all.py:334-427).  A few workgroups
per destination rank move that rank's contiguous token rows (tokens are pre-sorted by expert), the
per-expert splits and the optional fp32 scales straight from the caller's tensors into the peer's
receive buffer over xGMI (no staging copy), then raise a 64-bit signal; receive buffers are double-buffered by call parity
so the sender never waits for the receiver to have consumed the previous message.

On the GPU backend the call number lives in device memory: the kernel picks the half and the signal value itself and
copies what arrived into buffers whose address is the same every call, so ``fast_all_to_all`` (and
``all_to_all_post_process``, one compaction kernel, no host round trip) can be captured into a HIP graph and replayed.
"""
from __future__ import annotations

import ctypes
import dataclasses
import os
from typing import Any, List, Optional, Tuple

import torch

from .. import _lib
from ..utils.distributed import get_env, forbid_graph_capture

FP8_MAX = 448.0  # OCP e4m3fn


@dataclasses.dataclass(repr=False)
class AllToAllContext:
    max_m: int
    hidden: int
    rank: int
    num_tot_experts: int
    world_size: int
    experts_per_rank: int
    dtype: torch.dtype
    scale_dtype: torch.dtype
    scale_elems: int
    send_buf: torch.Tensor
    recv_buf: List[torch.Tensor]
    scale_send_buf: torch.Tensor
    scale_recv_buf: List[torch.Tensor]
    split_send_buf: torch.Tensor
    split_recv_buf: List[torch.Tensor]
    signal_buf: List[torch.Tensor]
    err: Optional[torch.Tensor]
    env: Any = None
    call_count: int = 1
    MOD_VALUE: int = 1000000
    # GPU backend: device call state and the parity-independent buffers fast_all_to_all returns views of
    state: Optional[torch.Tensor] = None
    out_rows: Optional[torch.Tensor] = None
    out_scale: Optional[torch.Tensor] = None
    out_splits: Optional[torch.Tensor] = None

    def check_error(self) -> None:
        code = int(self.err[0]) if self.err is not None else 0
        if code & 2:
            raise RuntimeError("fast_all_to_all: send_split_cumsum is inconsistent with the send tensor / max_m")
        if code:
            raise RuntimeError("fast_all_to_all: a device-side wait timed out")


def create_all_to_all_context(max_m, hidden, rank, num_tot_experts, WORLD_SIZE, experts_per_rank,
                              dtype=torch.bfloat16, scale_dtype=torch.float32, scale_group: int = 128,
                              env=None) -> AllToAllContext:
    env = env or get_env()
    heap = env.heap
    w = WORLD_SIZE
    scale_elems = max(1, hidden // scale_group)
    dev = env.device
    send_buf = torch.empty((max_m, hidden), dtype=dtype, device=dev)
    recv = heap.create_tensor_list([2, w * max_m, hidden], dtype, zero=False)
    scale_send = torch.empty((max_m, scale_elems), dtype=scale_dtype, device=dev)
    scale_recv = heap.create_tensor_list([2, w * max_m, scale_elems], scale_dtype, zero=False)
    split_send = torch.zeros((num_tot_experts + 1,), dtype=torch.int32, device=dev)
    split_recv = heap.create_tensor_list([2, w * experts_per_rank], torch.int32)
    signal = heap.create_tensor_list([2 * w], torch.int64, arena="sig")
    # [0] error word, [4 : 4 + w] per-destination slice counters (zero at rest: the kernel re-arms them)
    err = torch.zeros(4 + 16, dtype=torch.int32, device=dev) if env.backend == "hip" else None
    state = out_rows = out_scale = out_splits = None
    if env.backend == "hip":
        state = torch.zeros(4, dtype=torch.int32, device=dev)
        out_rows = torch.empty((w, max_m, hidden), dtype=dtype, device=dev)
        out_scale = torch.empty((w, max_m, scale_elems), dtype=scale_dtype, device=dev)
        out_splits = torch.zeros((w, experts_per_rank), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
    if env.cpu_group is not None:
        torch.distributed.barrier(group=env.cpu_group)
    return AllToAllContext(max_m, hidden, rank, num_tot_experts, w, experts_per_rank, dtype, scale_dtype, scale_elems,
                           send_buf, recv, scale_send, scale_recv, split_send, split_recv, signal, err, env,
                           state=state, out_rows=out_rows, out_scale=out_scale, out_splits=out_splits)


def fast_all_to_all(ctx: AllToAllContext, send_tensor: torch.Tensor, send_split_cumsum: torch.Tensor,
                    send_scale: Optional[torch.Tensor] = None):
    """send_tensor [num_tokens, hidden] sorted by destination expert; send_split_cumsum
    [num_tot_experts + 1] int32 exclusive cumsum.  Returns (recv_splits [w, EPR], recv [w, max_m,
    hidden], recv_scale or None): the context's result buffers on the GPU backend (same tensors every call, valid
    until the next call), views into this call's half of the receive buffers on the host backend."""
    env = ctx.env
    w, rank = ctx.world_size, ctx.rank
    n = send_tensor.shape[0]
    assert n <= ctx.max_m and send_tensor.shape[1] == ctx.hidden
    assert send_split_cumsum.numel() == ctx.num_tot_experts + 1
    half = ctx.call_count & 1
    with_scale = send_scale is not None
    if env.backend == "host":
        forbid_graph_capture("fast_all_to_all (host backend)", w)
        ctx.send_buf[:n].copy_(send_tensor)
        ctx.split_send_buf.copy_(send_split_cumsum.to(torch.int32))
        if with_scale:
            ctx.scale_send_buf[:n].copy_(send_scale)
        _a2a_host(ctx, n, half, with_scale)
    else:
        # The kernel pushes straight from the caller's tensors (stream order keeps them alive); the
        # context's staging buffers are only used when a tensor has to be converted first.
        assert send_tensor.dtype == ctx.dtype and send_tensor.is_cuda
        src = send_tensor if send_tensor.is_contiguous() else ctx.send_buf[:n].copy_(send_tensor)
        cum = send_split_cumsum
        if cum.dtype != torch.int32 or not cum.is_contiguous() or not cum.is_cuda:
            cum = ctx.split_send_buf.copy_(cum)
        scale = None
        if with_scale:
            assert send_scale.shape == (n, ctx.scale_elems)
            scale = send_scale
            if scale.dtype != ctx.scale_dtype or not scale.is_contiguous():
                scale = ctx.scale_send_buf[:n].copy_(send_scale)
        cs = env.heap.ctx_struct()
        row_bytes = ctx.hidden * src.element_size()
        _lib.check(
            _lib.hip().tdist_all_to_all(ctypes.byref(cs), _lib.ptr(src), _lib.ptr(ctx.recv_buf[rank]),
                                        _lib.ptr(scale) if with_scale else ctypes.c_void_p(0),
                                        _lib.ptr(ctx.scale_recv_buf[rank]), _lib.ptr(cum),
                                        _lib.ptr(ctx.split_recv_buf[rank]), _lib.ptr(ctx.signal_buf[rank]),
                                        _lib.ptr(ctx.err), ctypes.c_longlong(row_bytes), n, ctx.max_m,
                                        ctx.experts_per_rank, ctx.scale_elems, _a2a_slices(n * row_bytes, w, env.shared_gpu),
                                        ctypes.c_ulonglong(0), _lib.stream_handle(), _lib.ptr(ctx.state),
                                        _lib.ptr(ctx.out_rows), _lib.ptr(ctx.out_scale), _lib.ptr(ctx.out_splits)),
            "tdist_all_to_all")
        return ctx.out_splits, ctx.out_rows, (ctx.out_scale if with_scale else None)
    ctx.call_count = (ctx.call_count + 1) % ctx.MOD_VALUE or 1
    recv = ctx.recv_buf[rank][half].view(w, ctx.max_m, ctx.hidden)
    splits = ctx.split_recv_buf[rank][half].view(w, ctx.experts_per_rank)
    scale = ctx.scale_recv_buf[rank][half].view(w, ctx.max_m, ctx.scale_elems) if with_scale else None
    return splits, recv, scale


def _a2a_slices(total_bytes: int, w: int, shared_gpu: bool = False) -> int:
    """Workgroups per destination: ~128 KiB each, at most 4 (measured on MI355X for the 900 KiB
    per destination of the DeepSeek decode shape: 1 / 2 / 4 / 8 / 16 workgroups per destination
    take 43 / 32 / 26-30 / 34 / 45 us - more 1024-thread workgroups cost more to launch than they
    gain), and never more than 128 workgroups in the launch.  ``TDIST_A2A_SLICES`` overrides.  When all ranks share one GPU (virtual ranks) the 128 are
    shared too: every rank's waiting workgroups must be resident together with every producer."""
    forced = int(os.environ.get("TDIST_A2A_SLICES", "0"))
    if forced:
        return max(1, min(32, forced))
    w = max(w, 1)
    per_dest = max(1, total_bytes // w)
    budget = 128 // w if shared_gpu else 128
    return max(1, min(4, budget // w, per_dest // 131072))


def _a2a_host(ctx, n, half, with_scale):
    lib = _lib.host()
    w, rank, epr = ctx.world_size, ctx.rank, ctx.experts_per_rank
    cum = ctx.split_send_buf
    for i in range(w):
        d = (rank + i) % w
        st, ed = int(cum[d * epr]), int(cum[(d + 1) * epr])
        ctx.recv_buf[d][half, rank * ctx.max_m: rank * ctx.max_m + (ed - st)].copy_(ctx.send_buf[st:ed])
        if with_scale:
            ctx.scale_recv_buf[d][half, rank * ctx.max_m: rank * ctx.max_m + (ed - st)].copy_(ctx.scale_send_buf[st:ed])
        ctx.split_recv_buf[d][half, rank * epr:(rank + 1) * epr].copy_(cum[d * epr + 1:(d + 1) * epr + 1] - cum[d * epr:(d + 1) * epr])
        lib.tdh_notify64(ctypes.c_void_p(ctx.signal_buf[d].data_ptr() + 8 * (half * w + rank)), ctx.call_count, 1)
    if lib.tdh_wait64(ctypes.c_void_p(ctx.signal_buf[rank].data_ptr() + 8 * half * w), w, 0, ctx.call_count,
                      ctypes.c_int64(60000)) != 0:
        raise RuntimeError("fast_all_to_all(host) timed out")


def _compact_device(ctx, splits, recv, scale):
    w = ctx.world_size
    assert splits.dtype == torch.int32 and splits.is_contiguous() and recv.is_contiguous()
    out = torch.empty((w * ctx.max_m, ctx.hidden), dtype=recv.dtype, device=recv.device)
    out_scale = torch.empty((w * ctx.max_m, ctx.scale_elems), dtype=torch.float32, device=recv.device) if scale is not None else None
    counts = torch.empty(ctx.experts_per_rank + 1, dtype=torch.int32, device=recv.device)
    if scale is not None:
        assert scale.dtype == torch.float32 and scale.is_contiguous()
    _lib.check(
        _lib.hip().tdist_a2a_compact(_lib.ptr(recv), _lib.ptr(scale), _lib.ptr(splits), _lib.ptr(out), _lib.ptr(out_scale),
                                     _lib.ptr(counts), _lib.ptr(counts[ctx.experts_per_rank:]),
                                     ctypes.c_longlong(ctx.hidden * recv.element_size()), w, ctx.max_m,
                                     ctx.experts_per_rank, ctx.scale_elems, _lib.stream_handle()), "tdist_a2a_compact")
    return out, out_scale, counts[: ctx.experts_per_rank], counts[ctx.experts_per_rank:]


def all_to_all_post_process(ctx: AllToAllContext, splits: torch.Tensor, recv: torch.Tensor,
                            scale: Optional[torch.Tensor] = None, padded: bool = False):
    """Compact the per-source segments: returns (tokens [sum, hidden], scales or None,
    tokens_per_local_expert [EPR]).  ``padded=True`` (implied while a HIP graph is being captured) keeps everything
    on the device: tokens / scales come back with ``w * max_m`` rows, of which the first ``total`` are valid, and a
    fourth value ``total`` (int32 [1]) is returned."""
    w = ctx.world_size
    capturing = recv.is_cuda and torch.cuda.is_current_stream_capturing()
    if recv.is_cuda:
        out, out_scale, per_expert, total = _compact_device(ctx, splits, recv, scale)
        if padded or capturing:
            return out, out_scale, per_expert, total
        n = int(total)                      # the one host read an exactly sized result needs
        return out[:n], (out_scale[:n] if out_scale is not None else None), per_expert
    assert not padded, "padded compaction runs on the GPU backend"
    per_src = splits.sum(dim=1).tolist()
    rows = [recv[s, : per_src[s]] for s in range(w)]
    out = torch.cat(rows, dim=0)
    out_scale = torch.cat([scale[s, : per_src[s]] for s in range(w)], dim=0) if scale is not None else None
    return out, out_scale, splits.sum(dim=0)


# ---- fp8 group quantisation (per 128-element group, scale = max|x| / 448) --------------------
def quantize_fp8_groupwise(x: torch.Tensor, group: int = 128) -> Tuple[torch.Tensor, torch.Tensor]:
    M, H = x.shape
    assert H % group == 0
    if x.is_cuda:
        from .elementwise import fp8_group_quant

        return fp8_group_quant(x, group)
    xg = x.float().view(M, H // group, group)
    scale = xg.abs().amax(dim=-1).clamp_min(1e-12) / FP8_MAX
    q = (xg / scale.unsqueeze(-1)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).view(M, H)
    return q, scale


def dequantize_fp8_groupwise(q: torch.Tensor, scale: torch.Tensor, group: int = 128,
                             dtype=torch.bfloat16) -> torch.Tensor:
    M, H = q.shape
    return (q.float().view(M, H // group, group) * scale.float().unsqueeze(-1)).view(M, H).to(dtype)
