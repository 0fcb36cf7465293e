"""Expert-parallel dispatch / combine ("normal mode", large batches) inside one xGMI node.

API of the reference's ``EPAll2AllLayer`` (reference: python/triton_dist/layers/nvidia/ep_a2a_layer.py
:42-248) and helpers of kernels/nvidia/ep_a2a.py (``bincount`` :326-334, splits / offsets
:243-306, ``kernel_dispatch_token`` :36-148, ``kernel_combine_token`` :151-240).  Single node =>
the inter-node stage of the reference disappears:

  dispatch: (1) per-expert counts are exchanged with one small kernel; every rank then knows the
            full [src_rank, expert] table and computes, for each expert, where ITS tokens start in
            the owner's receive buffer (segments ordered by local expert, then source rank);
            (2) one wave per (token, k) allocates a row from a local per-expert cursor and stores the
            token row straight into the owner's buffer over xGMI; (3) a per-source flag tells the
            owner that all rows of that source have landed.
  combine : pull-based - for each token, the topk expert outputs are read from their owners'
            symmetric buffers at the recorded rows, weighted and summed.

Launches.  A dispatch is two kernels (counts + offsets, token push), a combine is one.  The epoch of a call lives
in device memory and every counter is re-armed by the kernel that used it, so launch arguments never change and a
dispatch -> experts -> combine round trip can be captured into a HIP graph: under capture (or with
``padded=True``) ``dispatch`` returns the whole receive buffer [cap_rows, hidden] with the row count on the device
(``num_recv_dev``, ``recv_per_expert``); eagerly it reads the count from a pinned host word and returns an exactly
sized view.  Experts may write their outputs straight into ``expert_out_buffer()`` - ``combine`` then pulls from it
without a staging copy.

Buffer reuse.  The counts exchange at the head of every dispatch is a full handshake: a rank leaves it only
after every peer's stream has reached the same dispatch, i.e. after the peer has finished (in stream order)
everything it launched for the previous dispatch / combine.  So the receive buffer and the expert-output buffer
are single: rows of dispatch n are overwritten by dispatch n + 1 only after every rank has enqueued it behind its
own consumers of dispatch n.  A consumer on ANOTHER stream has to be ordered before the next dispatch with an
event, as for any tensor.  What no buffering can make meaningful is a second combine for the same dispatch (its
flags already carry the epoch, nothing would be waited for): eager ``combine`` refuses it.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from .. import _lib
from ..utils.distributed import get_env


def bincount(x: torch.Tensor, length: int) -> torch.Tensor:
    return torch.bincount(x.reshape(-1).to(torch.int64), minlength=length).to(torch.int32)


def dispatch_offsets(table: torch.Tensor, world_size: int, experts_per_rank: int, rank: int):
    """Reference computation of what the counts kernel derives on the device.  table [w, E] (tokens of src rank s
    for expert e).  Returns
    dst_base [E]  - first row of (expert e, src=rank) inside the owner's receive buffer,
    recv_per_expert [EPR] - rows per local expert on this rank, num_recv (0-dim tensor)."""
    w, E = table.shape
    t = table.view(w, world_size, experts_per_rank).permute(1, 2, 0).contiguous()   # [owner, e_local, src]
    flat = t.view(world_size, -1).to(torch.int64)
    starts = torch.cumsum(flat, dim=1) - flat                                        # exclusive, per owner
    dst_base = starts.view(world_size, experts_per_rank, w)[:, :, rank].reshape(E).to(torch.int32)
    recv_per_expert = t[rank].sum(dim=1).to(torch.int32)
    return dst_base, recv_per_expert, flat[rank].sum()


class EPAll2AllLayer(torch.nn.Module):
    def __init__(self, ep_group, max_tokens: int, hidden: int, topk: int, rank: int, num_tot_experts: int,
                 local_world_size: int, world_size: int, dtype=torch.bfloat16, num_sm: int = 20,
                 capacity_factor: float = 2.0, env=None):
        super().__init__()
        self.env = env or get_env()
        assert world_size == local_world_size, "single xGMI node"
        assert num_tot_experts % world_size == 0
        assert self.env.backend == "hip", "EP dispatch runs on the GPU backend"
        self.max_tokens, self.hidden, self.topk, self.rank = max_tokens, hidden, topk, rank
        self.E, self.w, self.dtype, self.num_sm = num_tot_experts, world_size, dtype, num_sm
        self.epr = num_tot_experts // world_size
        self.cap_rows = int(max_tokens * topk * capacity_factor)
        heap = self.env.heap
        self.recv_buf = heap.create_tensor_list([self.cap_rows, hidden], dtype, zero=False)
        self.expert_out = heap.create_tensor_list([self.cap_rows, hidden], dtype, zero=False)
        self.table = heap.create_tensor_list([world_size, num_tot_experts], torch.int32)
        self.flags = heap.create_tensor_list([3 * world_size], torch.int32, arena="sig")
        dev = self.env.device
        E = self.E
        # [0:E] dispatch cursors, [E] dispatch done counter, [E+8] error word, [E+12] epoch, [E+13] counts-kernel arrivals
        self.scratch = torch.zeros(E + 16, dtype=torch.int32, device=dev)
        self.dst_base = torch.zeros(E, dtype=torch.int32, device=dev)
        self.recv_per_expert = torch.zeros(self.epr, dtype=torch.int32, device=dev)
        self.num_recv_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._num_recv_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._idx = torch.zeros((max_tokens, topk), dtype=torch.int32, device=dev)
        self._scatter = torch.full((max_tokens, topk), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        if self.env.cpu_group is not None:
            torch.distributed.barrier(group=self.env.cpu_group)
        self.scatter_idx = None
        self.exp_indices = None
        self.num_recv = 0          # rows of the latest eager dispatch (cap_rows after a padded one)
        self._combined = True      # the latest dispatch has been combined (or there has been none yet)

    # scratch slices
    def _err(self):
        return self.scratch[self.E + 8:]

    def _state(self):
        return self.scratch[self.E + 12:]

    def check_error(self):
        code = int(self.scratch[self.E + 8])
        if code & 2:
            raise RuntimeError("EPAll2AllLayer: rows were dropped - the receive capacity (capacity_factor) is too small")
        if code:
            raise RuntimeError("EPAll2AllLayer: a device-side wait timed out")

    def expert_out_buffer(self) -> torch.Tensor:
        """This rank's symmetric expert-output buffer [cap_rows, hidden]: write expert results for received row i into
        row i and hand (a leading slice of) it to ``combine`` - no staging copy is made then."""
        return self.expert_out[self.rank]

    def preprocess(self, exp_indices: torch.Tensor):
        """Counts exchange + offsets (one launch); returns device tensors (dst_base [E], recv_per_expert [EPR],
        num_recv [1])."""
        lib, cs = _lib.hip(), self.env.heap.ctx_struct()
        idx = exp_indices
        if idx.dtype != torch.int32 or not idx.is_contiguous():
            idx = idx.to(torch.int32).contiguous()
        _lib.check(
            lib.tdist_ep_exchange_counts(ctypes.byref(cs), _lib.ptr(idx), idx.numel(), _lib.ptr(self.table[self.rank]),
                                         _lib.ptr(self.flags[self.rank]), _lib.ptr(self._err()), _lib.ptr(self._state()),
                                         _lib.ptr(self.dst_base), _lib.ptr(self.recv_per_expert),
                                         _lib.ptr(self.num_recv_dev), _lib.ptr(self._num_recv_host), _lib.ptr(self.scratch),
                                         self.E, self.epr, _lib.stream_handle()), "ep_exchange_counts")
        return self.dst_base, self.recv_per_expert, self.num_recv_dev

    def dispatch(self, input: torch.Tensor, exp_indices: torch.Tensor, padded: bool = False) -> torch.Tensor:
        """input [T, hidden], exp_indices [T, topk] -> tokens received for this rank's experts, grouped by local expert
        (``self.recv_per_expert`` rows each): [num_recv, hidden] eagerly, the whole [cap_rows, hidden] buffer (first
        ``num_recv_dev`` rows valid) with ``padded=True`` or while a HIP graph is being captured."""
        T = input.shape[0]
        assert T <= self.max_tokens and input.shape[1] == self.hidden and input.dtype == self.dtype
        assert tuple(exp_indices.shape) == (T, self.topk)
        padded = padded or torch.cuda.is_current_stream_capturing()
        idx = self._idx[:T]
        idx.copy_(exp_indices)
        self.preprocess(idx)
        self.scatter_idx = self._scatter[:T]
        self.exp_indices = idx
        self._combined = False
        recv = self.recv_buf[self.rank]
        src = input if input.is_contiguous() else input.contiguous()
        lib, cs = _lib.hip(), self.env.heap.ctx_struct()
        w = self.w
        _lib.check(
            lib.tdist_ep_dispatch(ctypes.byref(cs), _lib.ptr(src), _lib.ptr(idx), _lib.ptr(self.dst_base),
                                  _lib.ptr(self.scratch), _lib.ptr(recv), _lib.ptr(self.scatter_idx),
                                  _lib.ptr(self.flags[self.rank][w:]), _lib.ptr(self.scratch[self.E:]), _lib.ptr(self._err()),
                                  _lib.ptr(self._state()), ctypes.c_longlong(self.hidden * input.element_size()), T,
                                  self.topk, self.E, self.epr, self.cap_rows, self.num_sm * 2, _lib.stream_handle()),
            "ep_dispatch")
        if padded:
            self.num_recv = self.cap_rows
            return recv
        torch.cuda.current_stream().synchronize()          # the row count sizes the returned view
        self.num_recv = int(self._num_recv_host[0])
        if self.num_recv > self.cap_rows:
            raise RuntimeError(f"EP dispatch: {self.num_recv} rows exceed the receive capacity {self.cap_rows}")
        return recv[: self.num_recv]

    def combine(self, input: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        """input [num_recv (or more), hidden] = expert outputs in dispatch order -> [T, hidden].  A tensor that starts
        at ``expert_out_buffer()`` is used in place."""
        capturing = torch.cuda.is_current_stream_capturing()
        if self._combined and not capturing:
            raise RuntimeError("EPAll2AllLayer.combine: no dispatch since the last combine (one combine per dispatch)")
        assert input.shape[1] == self.hidden and input.dtype == self.dtype and input.shape[0] <= self.cap_rows
        assert input.shape[0] >= min(self.num_recv, self.cap_rows) or capturing
        self._combined = True
        T = self.exp_indices.shape[0]
        expert_out = self.expert_out[self.rank]
        if input.data_ptr() != expert_out.data_ptr() or not input.is_contiguous():
            expert_out[: input.shape[0]].copy_(input)
        out = torch.empty((T, self.hidden), dtype=self.dtype, device=input.device)
        wts = None
        if weights is not None:
            wts = weights if (weights.dtype == torch.float32 and weights.is_contiguous()) else weights.to(torch.float32).contiguous()
        lib, cs = _lib.hip(), self.env.heap.ctx_struct()
        _lib.check(
            lib.tdist_ep_combine(ctypes.byref(cs), _lib.ptr(expert_out), _lib.ptr(self.exp_indices),
                                 _lib.ptr(self.scatter_idx), _lib.ptr(wts), _lib.ptr(out),
                                 _lib.ptr(self.flags[self.rank][2 * self.w:]), _lib.ptr(self._err()), _lib.ptr(self._state()), T,
                                 self.topk, self.epr, self.hidden, self.num_sm * 4, _lib.stream_handle()),
            "ep_combine")
        return out
