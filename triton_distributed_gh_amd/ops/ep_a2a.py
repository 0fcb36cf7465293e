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

Buffer reuse.  The counts exchange at the head of every dispatch is a full handshake: a rank leaves it only
after every peer's stream has reached the same dispatch, i.e. after the peer has finished (in stream order)
everything it launched for the previous dispatch / combine.  On top of that the receive buffer and the
expert-output buffer are double-buffered by call parity, so the rows handed out by dispatch n stay intact
while dispatch n + 1 is in flight - a consumer of dispatch n on another stream, or a peer still pulling
expert outputs of combine n, never sees rows of the next call.  What no buffering can make meaningful is a
second combine for the same dispatch (its flags already carry the epoch, nothing would be waited for):
``combine`` refuses it.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from .. import _lib
from ..utils.distributed import forbid_graph_capture, get_env


def bincount(x: torch.Tensor, length: int) -> torch.Tensor:
    return torch.bincount(x.reshape(-1).to(torch.int64), minlength=length).to(torch.int32)


def dispatch_offsets(table: torch.Tensor, world_size: int, experts_per_rank: int, rank: int):
    """table [w, E] (tokens of src rank s for expert e).  Returns
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
        self.max_tokens, self.hidden, self.topk, self.rank = max_tokens, hidden, topk, rank
        self.E, self.w, self.dtype, self.num_sm = num_tot_experts, world_size, dtype, num_sm
        self.epr = num_tot_experts // world_size
        self.cap_rows = int(max_tokens * topk * capacity_factor)
        heap = self.env.heap
        self.recv_buf = heap.create_tensor_list([2, self.cap_rows, hidden], dtype, zero=False)     # [parity]
        self.expert_out = heap.create_tensor_list([2, self.cap_rows, hidden], dtype, zero=False)   # [parity]
        self.table = heap.create_tensor_list([world_size, num_tot_experts], torch.int32)
        self.flags = heap.create_tensor_list([3 * world_size], torch.int32, arena="sig")
        dev = self.env.device
        if self.env.backend == "hip":
            self.scratch = torch.zeros(num_tot_experts + 16, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
        if self.env.cpu_group is not None:
            torch.distributed.barrier(group=self.env.cpu_group)
        self.epoch = 0
        self.scatter_idx = None
        self.exp_indices = None
        self.recv_per_expert = None
        self.num_recv = 0
        self._combined = True      # the latest dispatch has been combined (or there has been none yet)

    def check_error(self):
        if int(self.scratch[self.E + 8]) != 0:
            raise RuntimeError("EPAll2AllLayer: a device-side wait timed out")

    def preprocess(self, exp_indices: torch.Tensor):
        """Counts exchange; returns (dst_base, recv_per_expert, num_recv)."""
        lib, cs = _lib.hip(), self.env.heap.ctx_struct()
        counts = bincount(exp_indices, self.E)
        w = self.w
        _lib.check(lib.tdist_ep_exchange_counts(ctypes.byref(cs), _lib.ptr(counts), _lib.ptr(self.table[self.rank]),
                                                _lib.ptr(self.flags[self.rank]), _lib.ptr(self.scratch[self.E + 8:]),
                                                self.E, self.epoch, _lib.stream_handle()), "ep_exchange_counts")
        return dispatch_offsets(self.table[self.rank], w, self.epr, self.rank)

    def dispatch(self, input: torch.Tensor, exp_indices: torch.Tensor) -> torch.Tensor:
        """input [T, hidden], exp_indices [T, topk] -> tokens received for this rank's experts
        [num_recv, hidden], grouped by local expert (``self.recv_per_expert`` rows each)."""
        assert self.env.backend == "hip", "EP dispatch runs on the GPU backend"
        T = input.shape[0]
        assert T <= self.max_tokens and input.shape[1] == self.hidden and input.dtype == self.dtype
        # (the receive count crosses to the host to size the output, and epoch / buffer parity are host state)
        forbid_graph_capture("EPAll2AllLayer.dispatch", self.w)
        self.epoch += 1
        idx = exp_indices.to(torch.int32).contiguous()
        dst_base, self.recv_per_expert, num_recv = self.preprocess(idx)
        self.num_recv = int(num_recv)                      # host handshake: sizes the output view
        if self.num_recv > self.cap_rows:
            raise RuntimeError(f"EP dispatch: {self.num_recv} rows exceed the receive capacity {self.cap_rows}")
        self.scatter_idx = torch.empty((T, self.topk), dtype=torch.int32, device=input.device)
        self.exp_indices = idx
        self._combined = False
        recv = self.recv_buf[self.rank][self.epoch & 1]
        lib, cs = _lib.hip(), self.env.heap.ctx_struct()
        w = self.w
        _lib.check(
            lib.tdist_ep_dispatch(ctypes.byref(cs), _lib.ptr(input.contiguous()), _lib.ptr(idx), _lib.ptr(dst_base),
                                  _lib.ptr(self.scratch), _lib.ptr(recv), _lib.ptr(self.scatter_idx),
                                  _lib.ptr(self.flags[self.rank][w:]), _lib.ptr(self.scratch[self.E:]),
                                  _lib.ptr(self.scratch[self.E + 8:]),
                                  ctypes.c_longlong(self.hidden * input.element_size()), T, self.topk, self.E, self.epr,
                                  self.epoch, self.num_sm * 2, _lib.stream_handle()), "ep_dispatch")
        return recv[: self.num_recv]

    def combine(self, input: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        """input [num_recv, hidden] = expert outputs in dispatch order -> [T, hidden]."""
        if self._combined:
            raise RuntimeError("EPAll2AllLayer.combine: no dispatch since the last combine (one combine per dispatch)")
        assert input.shape[0] == self.num_recv
        self._combined = True
        T = self.exp_indices.shape[0]
        expert_out = self.expert_out[self.rank][self.epoch & 1]
        expert_out[: self.num_recv].copy_(input)
        out = torch.empty((T, self.hidden), dtype=self.dtype, device=input.device)
        wts = weights.to(torch.float32).contiguous() if weights is not None else None
        lib, cs = _lib.hip(), self.env.heap.ctx_struct()
        _lib.check(
            lib.tdist_ep_combine(ctypes.byref(cs), _lib.ptr(expert_out), _lib.ptr(self.exp_indices),
                                 _lib.ptr(self.scatter_idx), _lib.ptr(wts), _lib.ptr(out),
                                 _lib.ptr(self.flags[self.rank][2 * self.w:]), _lib.ptr(self.scratch[self.E + 8:]), T,
                                 self.topk, self.epr, self.hidden, self.epoch, self.num_sm * 4, _lib.stream_handle()),
            "ep_combine")
        return out
