"""Node-wide barrier and stream-ordered signals.

* ``barrier_all_on_stream`` / ``barrier_all_ipc`` - slot-matrix barrier kernel (one wave, lane i
  serves peer i) over a symmetric int32[num_ranks] buffer (reference:
  python/triton_dist/kernels/amd/common_ops.py:45-69).
* ``wait_eq`` / ``set_signal`` - ``hipStreamWaitValue32/64`` / ``hipStreamWriteValue32/64`` on a
  stream (reference: kernels/amd/common_ops.py:72-110, kernels/common_ops.py).
* ``BarrierAllContext`` - owns the symmetric slot buffer (reference: kernels/nvidia/common_ops.py
  :212-243).
On the host backend the same calls run through the CPU runtime.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from .. import _lib
from ..utils.distributed import get_env


class BarrierAllContext:
    """Collective constructor: allocates the barrier's symmetric slots in the signal arena."""

    def __init__(self, env=None):
        self.env = env or get_env()
        self.slots = self.env.heap.create_tensor_list([self.env.world_size], torch.int32, arena="sig")
        self.ok = (torch.ones(1, dtype=torch.int32, device=self.env.device) if self.env.backend == "hip" else None)
        if self.env.backend == "hip":
            torch.cuda.synchronize()
        if self.env.cpu_group is not None:
            torch.distributed.barrier(group=self.env.cpu_group)


def barrier_all_on_stream(ctx: BarrierAllContext, stream: Optional["torch.cuda.Stream"] = None) -> None:
    env = ctx.env
    if env.backend == "host":
        lib = _lib.host()
        ptrs = (ctypes.c_void_p * env.world_size)(*[t.data_ptr() for t in ctx.slots])
        rc = lib.tdh_barrier_all(ptrs, env.rank, env.world_size, ctypes.c_int64(60000))
        if rc != 0:
            raise RuntimeError("barrier_all (host) timed out")
        return
    cs = env.heap.ctx_struct()
    _lib.check(_lib.hip().tdist_barrier_all(ctypes.byref(cs), _lib.ptr(ctx.slots[env.rank]), _lib.ptr(ctx.ok),
                                            _lib.stream_handle(stream)), "tdist_barrier_all")


# name used by the reference's AMD kernels
barrier_all_ipc = barrier_all_on_stream


def set_signal(signal: torch.Tensor | int, value: int, stream=None, bits: Optional[int] = None) -> None:
    """Stream-ordered store of ``value`` to a signal word (tensor element or raw device address)."""
    addr = signal if isinstance(signal, int) else signal.data_ptr()
    if bits is None:
        bits = 64 if (not isinstance(signal, int) and signal.element_size() == 8) else 32
    if not torch.cuda.is_available():
        lib = _lib.host()
        (lib.tdh_notify64 if bits == 64 else lib.tdh_notify32)(ctypes.c_void_p(addr), value, 1)
        return
    lib = _lib.hip()
    if bits == 64:
        _lib.check(lib.tdist_stream_write_value64(_lib.stream_handle(stream), ctypes.c_void_p(addr),
                                                  ctypes.c_ulonglong(value)), "hipStreamWriteValue64")
    else:
        _lib.check(lib.tdist_stream_write_value32(_lib.stream_handle(stream), ctypes.c_void_p(addr),
                                                  ctypes.c_uint(value)), "hipStreamWriteValue32")


def wait_eq(signal: torch.Tensor | int, value: int, stream=None, bits: Optional[int] = None,
            timeout_ms: int = 60000, ge: bool = False) -> None:
    """Make ``stream`` wait until the signal word equals (``ge``: is at least) ``value``."""
    addr = signal if isinstance(signal, int) else signal.data_ptr()
    if bits is None:
        bits = 64 if (not isinstance(signal, int) and signal.element_size() == 8) else 32
    if not torch.cuda.is_available():
        lib = _lib.host()
        fn = lib.tdh_wait64 if bits == 64 else lib.tdh_wait32
        if fn(ctypes.c_void_p(addr), 1, 5 if ge else 0, value, ctypes.c_int64(timeout_ms)) != 0:
            raise RuntimeError("wait_eq (host) timed out")
        return
    lib = _lib.hip()
    if bits == 64:
        _lib.check(lib.tdist_stream_wait_value64(_lib.stream_handle(stream), ctypes.c_void_p(addr),
                                                 ctypes.c_ulonglong(value), 1 if ge else 0), "hipStreamWaitValue64")
    else:
        _lib.check(lib.tdist_stream_wait_value32(_lib.stream_handle(stream), ctypes.c_void_p(addr),
                                                 ctypes.c_uint(value), 1 if ge else 0), "hipStreamWaitValue32")


def memcpy_async(dst: torch.Tensor | int, src: torch.Tensor | int, nbytes: int, stream=None,
                 use_copy_engine: bool = True) -> None:
    d = dst if isinstance(dst, int) else dst.data_ptr()
    s = src if isinstance(src, int) else src.data_ptr()
    _lib.check(_lib.hip().tdist_memcpy_async(ctypes.c_void_p(d), ctypes.c_void_p(s), ctypes.c_size_t(nbytes),
                                             1 if use_copy_engine else 0, _lib.stream_handle(stream)), "memcpy_async")


def barrier_on_this_grid_smoke(num_blocks: int = 256, rounds: int = 8, stream=None):
    """Runs the device-side grid barrier (csrc/include/tdist/grid_barrier.hpp - the counterpart of
    the reference's ``barrier_on_this_grid`` device function, kernels/nvidia/common_ops.py:56-130)
    ``2 * rounds`` times over ``num_blocks`` resident workgroups.  Returns (stale_reads, timeouts)."""
    import ctypes

    from .. import _lib
    dev = torch.device("cuda", torch.cuda.current_device())
    state = torch.zeros(4, dtype=torch.int32, device=dev)
    words = torch.zeros(num_blocks, dtype=torch.int32, device=dev)
    bad = torch.zeros(2, dtype=torch.int32, device=dev)
    s = stream or torch.cuda.current_stream()
    _lib.check(_lib.hip().tdist_test_grid_barrier(_lib.ptr(state), _lib.ptr(words), rounds, _lib.ptr(bad), num_blocks,
                                                  _lib.stream_handle(s)), "tdist_test_grid_barrier")
    s.synchronize()
    return int(bad[0]), int(bad[1])


# ---- small host helpers the reference keeps next to its device helpers (kernels/nvidia/common_ops.py) ----
def next_power_of_2(x: int) -> int:
    return 1 if x <= 1 else 1 << (int(x) - 1).bit_length()


def log2(x: int) -> int:
    """Exact log2 of a power of two."""
    x = int(x)
    if x <= 0 or x & (x - 1):
        raise ValueError(f"log2: {x} is not a power of two")
    return x.bit_length() - 1


# The reference's node barrier has several device forms (atomic-CAS per block, non-atomic slot matrix,
# ``barrier_all_intra_node_*``, kernels/nvidia/common_ops.py:95-243); here they are ONE kernel - every rank
# writes its slot on every peer and waits for all of its own (csrc/include/tdist/shmem_device.hpp:
# barrier_all_block) - launched on a stream.
barrier_all_intra_node_non_atomic = barrier_all_intra_node_atomic_cas_block = barrier_all_on_stream
barrier_all_intra_node_non_atomic_block = barrier_all_on_stream
