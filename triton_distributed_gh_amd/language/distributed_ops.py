"""wait / consume_token / notify / symm_at / rank / num_ranks on symmetric tensors.

Signatures follow reference python/triton_dist/language/distributed_ops.py:56-111:
``wait(barrierPtrs, numBarriers, scope, semantic, waitValue=1)``, ``consume_token(value, token)``,
``rank(axis)``, ``num_ranks(axis)``, ``symm_at(ptr, rank)``,
``notify(ptr, rank, signal=1, sig_op="set", comm_scope="intra_node")``.
"""
from __future__ import annotations

import ctypes
import enum
from typing import Optional

import torch

from .. import _lib
from ..utils.distributed import get_env


class SIGNAL_OP(enum.IntEnum):
    SET = 1
    ADD = 2


class COMM_SCOPE(enum.IntEnum):
    GPU = 1
    INTRA_NODE = 2
    INTER_NODE = 3


_SCOPES = {"workgroup": 0, "cta": 0, "gpu": 1, "agent": 1, "sys": 2, "system": 2}
_SIG = {"set": SIGNAL_OP.SET, "add": SIGNAL_OP.ADD}
_COMM = {"gpu": COMM_SCOPE.GPU, "intra_node": COMM_SCOPE.INTRA_NODE, "inter_node": COMM_SCOPE.INTER_NODE}


def rank(axis: int = -1, env=None) -> int:
    return (env or get_env()).rank


def num_ranks(axis: int = -1, env=None) -> int:
    return (env or get_env()).world_size


def symm_at(ptr: torch.Tensor, rank: int, env=None) -> torch.Tensor:
    """View of symmetric tensor ``ptr`` as it exists on ``rank`` (same offset in that rank's heap)."""
    return (env or get_env()).heap.peer_view(ptr, rank)


def consume_token(value, token):
    """Identity: ties ``value`` to the wait that produced ``token``.  On a stream the order is the
    stream order; in device code it is an opaque asm use (primitives.hpp: consume_token)."""
    return value


def wait(barrierPtrs: torch.Tensor, numBarriers: int, scope: str = "sys", semantic: str = "acquire", waitValue: int = 1,
         stream=None, env=None, timeout_ms: int = 60000) -> int:
    """Block (the stream, or the calling thread on the host backend) until the first
    ``numBarriers`` words of ``barrierPtrs`` equal ``waitValue``.  Returns a token (0)."""
    env = env or get_env()
    if scope not in _SCOPES or semantic not in ("acquire", "relaxed"):
        raise ValueError(f"bad scope/semantic: {scope}/{semantic}")
    bits = barrierPtrs.element_size() * 8
    assert bits in (32, 64), "signals are int32 or int64"
    if env.backend == "host":
        fn = _lib.host().tdh_wait32 if bits == 32 else _lib.host().tdh_wait64
        ref = ctypes.c_int32(waitValue) if bits == 32 else ctypes.c_int64(waitValue)
        if fn(ctypes.c_void_p(barrierPtrs.data_ptr()), int(numBarriers), 0, ref, ctypes.c_int64(timeout_ms)) != 0:
            raise TimeoutError("wait: signal never arrived")
        return 0
    lib = _lib.hip()
    fn = lib.tdist_stream_wait_value32 if bits == 32 else lib.tdist_stream_wait_value64
    for i in range(int(numBarriers)):
        addr = barrierPtrs.data_ptr() + i * (bits // 8)
        val = ctypes.c_int32(waitValue) if bits == 32 else ctypes.c_int64(waitValue)
        _lib.check(fn(_lib.stream_handle(stream), ctypes.c_void_p(addr), val, 0), "stream wait value")
    return 0


def notify(ptr: torch.Tensor, rank: int, signal: int = 1, sig_op: str = "set", comm_scope: str = "intra_node",
           stream=None, env=None) -> None:
    """Set / add ``signal`` on rank ``rank``'s copy of the symmetric word ``ptr`` after everything
    issued before it (stream order on the GPU backend, program order on the host backend)."""
    env = env or get_env()
    op, scope = _SIG[sig_op], _COMM[comm_scope]
    bits = ptr.element_size() * 8
    if env.backend == "host":
        dst = ptr if scope == COMM_SCOPE.GPU else env.heap.peer_view(ptr, rank)
        fn = _lib.host().tdh_notify32 if bits == 32 else _lib.host().tdh_notify64
        fn(ctypes.c_void_p(dst.data_ptr()), int(signal), int(op))
        return
    if bits == 32:
        cs = env.heap.ctx_struct()
        _lib.check(_lib.hip().tdist_test_notify(ctypes.byref(cs), _lib.ptr(ptr), int(rank), int(signal), int(op),
                                                int(scope), _lib.stream_handle(stream)), "notify")
        return
    assert op == SIGNAL_OP.SET, "64-bit signals support 'set' from the host API"
    addr = ptr.data_ptr() if scope == COMM_SCOPE.GPU else env.heap.peer_ptr(ptr, rank)
    _lib.check(_lib.hip().tdist_stream_write_value64(_lib.stream_handle(stream), ctypes.c_void_p(addr),
                                                     ctypes.c_int64(signal)), "stream write value")
