"""Host launchers of the primitive exercise kernels (csrc/ops/primitives_test.hip): ring queue,
flag wait / notify, SHMEM put / get / signal smoke and the scoped atomics - the scenarios of the
reference's primitive tests (test/amd/test_distributed-notify-wait.py, test_wait_and_notify.py,
test_rocshmem_api.py, test_libdevice.py)."""
from __future__ import annotations

import ctypes

import torch

from .. import _lib
from ..utils.distributed import get_env


def ring_queue(queue, signal, input, output, peer, queue_size, item_elems, num_items, num_producers=16, num_consumers=4,
               err=None, stream=None, env=None):
    """Producers push ``num_items`` items of ``item_elems`` int32 into ``peer``'s symmetric queue,
    consumers pop this rank's queue into ``output``; slot hand-over by notify / wait (signal value
    2r = free, 2r + 1 = full in round r)."""
    env = env or get_env()
    cs = env.heap.ctx_struct()
    _lib.check(_lib.hip().tdist_test_ring_queue(ctypes.byref(cs), _lib.ptr(queue), _lib.ptr(signal), _lib.ptr(input),
                                                _lib.ptr(output), int(peer), int(queue_size), int(item_elems),
                                                int(num_items), int(num_producers), int(num_consumers), _lib.ptr(err),
                                                _lib.stream_handle(stream)), "ring_queue")


def wait_flag(flag, value, scope=2, stream=None):
    """Launch a kernel that waits for ``flag == value``; returns the int64 [2] status tensor
    (status[0]: 1 seen / 2 timed out, status[1]: ticks waited)."""
    status = torch.zeros(2, dtype=torch.int64, device=flag.device)
    _lib.check(_lib.hip().tdist_test_wait_flag(_lib.ptr(flag), int(value), int(scope), _lib.ptr(status),
                                               _lib.stream_handle(stream)), "wait_flag")
    return status


def shmem_api_smoke(buf, sig, src, readback, n, sig_value, stream=None, env=None):
    env = env or get_env()
    cs = env.heap.ctx_struct()
    _lib.check(_lib.hip().tdist_test_shmem_api(ctypes.byref(cs), _lib.ptr(buf), _lib.ptr(sig), _lib.ptr(src),
                                               _lib.ptr(readback), int(n), ctypes.c_ulonglong(sig_value),
                                               _lib.stream_handle(stream)), "shmem_api")


def atomics_smoke(stream=None):
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = torch.zeros(8, dtype=torch.int32, device="cuda")
    _lib.check(_lib.hip().tdist_test_atomics(_lib.ptr(word), _lib.ptr(out), _lib.stream_handle(stream)), "atomics")
    return word, out


def language_extra_smoke(buf, sync, out, env=None, stream=None):
    """Run the language_extra test kernel (csrc/ops/primitives_test.hip) on this rank: ``buf`` symmetric uint4
    area of (2 w + 2) x 128 vectors (int32 [(2 w + 2) * 512]), ``sync`` symmetric int32 [4] (zero), ``out`` int32 [4]."""
    cs = (env or get_env()).heap.ctx_struct()
    _lib.check(_lib.hip().tdist_test_language_extra(ctypes.byref(cs), _lib.ptr(buf), _lib.ptr(sync), _lib.ptr(out),
                                                    _lib.stream_handle(stream)), "language_extra")


def ld_reduce_smoke(src, dst, env=None, stream=None):
    """dst[128 x 8 bf16] = sum over ranks of the symmetric ``src`` (multimem_ld_reduce_v4_bf16)."""
    cs = (env or get_env()).heap.ctx_struct()
    _lib.check(_lib.hip().tdist_test_ld_reduce(ctypes.byref(cs), _lib.ptr(src), _lib.ptr(dst), _lib.stream_handle(stream)),
               "ld_reduce")
