"""Point-to-point copies between ranks for pipeline parallelism
(reference: python/triton_dist/kernels/nvidia/p2p.py:30-85 - ``p2p_copy_kernel`` pulls a peer's
symmetric buffer into the local symmetric buffer, ``p2p_copy_remote_to_local_kernel`` straight into
a local tensor).  Both are the pull kernel of csrc/ops/comm.hip (wide loads over xGMI)."""
from __future__ import annotations

import ctypes

import torch

from .. import _lib
from ..utils.distributed import get_env


def p2p_copy_remote_to_local(local_dst: torch.Tensor, symm_src: torch.Tensor, peer: int, num_sms: int = 4, stream=None,
                             env=None) -> torch.Tensor:
    """local_dst <- peer's copy of symmetric tensor ``symm_src`` (first ``local_dst.nbytes`` bytes)."""
    env = env or get_env()
    nbytes = local_dst.numel() * local_dst.element_size()
    assert local_dst.is_contiguous() and nbytes <= symm_src.numel() * symm_src.element_size()
    if env.backend == "host":
        src = env.heap.peer_view(symm_src, peer).reshape(-1).view(torch.uint8)[:nbytes]
        local_dst.view(-1).view(torch.uint8).copy_(src)
        return local_dst
    assert nbytes % 16 == 0, "p2p copies move 16-byte vectors"
    cs = env.heap.ctx_struct()
    _lib.check(_lib.hip().tdist_p2p_pull(ctypes.byref(cs), _lib.ptr(local_dst), _lib.ptr(symm_src),
                                         ctypes.c_longlong(nbytes), int(peer), int(num_sms),
                                         _lib.stream_handle(stream)), "tdist_p2p_pull")
    return local_dst


def p2p_copy(symm_buf: torch.Tensor, peer: int, nbytes: int = -1, num_sms: int = 4, stream=None, env=None):
    """Own copy of ``symm_buf`` <- peer's copy (getmem into the local symmetric buffer)."""
    n = symm_buf.numel() * symm_buf.element_size() if nbytes < 0 else nbytes
    flat = symm_buf.view(-1).view(torch.uint8)[:n]
    return p2p_copy_remote_to_local(flat, symm_buf, peer, num_sms, stream, env)
