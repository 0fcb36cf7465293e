"""Pipeline-parallel point-to-point primitive: symmetric buffers + signal words.

Reference: layers/nvidia/p2p.py:40-132 (``CommOp``) and kernels/nvidia/p2p.py:30-85.  The sender
writes activations into its OWN symmetric buffer and raises a signal; the receiver waits on the
sender's signal, pulls the buffer over xGMI with a small copy kernel, and clears the signal.
Signals are 64-bit words in the fine-grained signal arena, written / awaited with stream memory
operations so no kernel launch or host sync is involved in the handshake.
"""
from __future__ import annotations

import ctypes

import torch

from .. import _lib
from ..ops.common_ops import set_signal, wait_eq
from ..utils.distributed import get_env


class CommOp(torch.nn.Module):

    def __init__(self, max_tokens, token_dim, pp_rank, pp_size, pp_group, dtype, num_buffers, workspace_size=128,
                 env=None):
        super().__init__()
        self.env = env or get_env()
        self.max_tokens, self.token_dim = max_tokens, token_dim
        self.pp_rank, self.pp_size, self.pp_group, self.dtype = pp_rank, pp_size, pp_group, dtype
        heap = self.env.heap
        self._comm_buffers = [heap.create_tensor([max_tokens, token_dim], dtype) for _ in range(num_buffers)]
        self._signals = [heap.create_tensor([1], torch.int64, arena="sig") for _ in range(num_buffers)]
        self._workspaces = [heap.create_tensor([workspace_size], torch.uint8) for _ in range(num_buffers)]
        if self.env.backend == "hip":
            torch.cuda.synchronize()
        if self.env.cpu_group is not None:
            torch.distributed.barrier(group=self.env.cpu_group)

    def _global_rank(self, pp_rank: int) -> int:
        if self.pp_group is not None and torch.distributed.is_initialized():
            return torch.distributed.get_global_rank(self.pp_group, pp_rank)
        return pp_rank

    def get_buffer(self, buffer_id):
        assert buffer_id < len(self._comm_buffers)
        return self._comm_buffers[buffer_id]

    def read(self, pp_rank, buffer_id, local_tensor, sm=4, stream=None, fused=False):
        """Pull ``local_tensor.nbytes`` bytes of rank ``pp_rank``'s buffer into ``local_tensor``."""
        assert buffer_id < len(self._comm_buffers)
        peer = self._global_rank(pp_rank)
        nbytes = local_tensor.numel() * local_tensor.element_size()
        assert nbytes <= self._comm_buffers[buffer_id].numel() * self._comm_buffers[buffer_id].element_size()
        if self.env.backend == "host":
            src = self.env.heap.peer_views(self._comm_buffers[buffer_id])[peer]
            local_tensor.view(-1).copy_(src.view(-1)[: local_tensor.numel()].to(local_tensor.dtype))
            return
        assert local_tensor.is_contiguous() and nbytes % 16 == 0
        cs = self.env.heap.ctx_struct()
        _lib.check(
            _lib.hip().tdist_p2p_pull(ctypes.byref(cs), _lib.ptr(local_tensor), _lib.ptr(self._comm_buffers[buffer_id]),
                                      ctypes.c_longlong(nbytes), peer, sm, _lib.stream_handle(stream)), "tdist_p2p_pull")

    def set_signal(self, pp_rank, buffer_id, value, stream=None):
        peer = self._global_rank(pp_rank)
        addr = self.env.heap.peer_ptr(self._signals[buffer_id], peer)
        set_signal(addr, value, stream, bits=64)

    def wait_signal(self, pp_rank, buffer_id, value, stream=None):
        peer = self._global_rank(pp_rank)
        addr = self.env.heap.peer_ptr(self._signals[buffer_id], peer)
        wait_eq(addr, value, stream, bits=64)

    def wait_local_signal(self, buffer_id, value, stream=None):
        self.wait_signal(self.pp_rank, buffer_id, value, stream)
