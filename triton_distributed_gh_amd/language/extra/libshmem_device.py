"""SHMEM-style API over the symmetric heap (host-callable).

Names and constants follow reference python/triton_dist/language/extra/libshmem_device.py:28-340.
The DEVICE implementations (thread / wave / workgroup cooperative puts and gets, signal ops, waits,
barriers) live in csrc/include/tdist/shmem_device.hpp and are called from HIP kernels; the
functions below drive the same symmetric memory from the host: copies run on the copy engines in
stream order, signals are stream memory operations.  Scope suffixes (``_block`` / ``_warp``) only
matter inside kernels and resolve to the same host operation here.
"""
from __future__ import annotations

import ctypes

import torch

from ... import _lib
from ...ops.common_ops import memcpy_async, set_signal, wait_eq
from ...utils.distributed import get_env
from ..distributed_ops import notify

NVSHMEM_CMP_EQ, NVSHMEM_CMP_NE, NVSHMEM_CMP_GT, NVSHMEM_CMP_LE, NVSHMEM_CMP_LT, NVSHMEM_CMP_GE = range(6)
NVSHMEM_SIGNAL_SET = 9
NVSHMEM_SIGNAL_ADD = 10
NVSHMEM_TEAM_INVALID = -1
NVSHMEM_TEAM_WORLD = 0
NVSHMEM_TEAM_SHARED = 1
NVSHMEMX_TEAM_NODE = 2


def my_pe():
    return get_env().rank


def n_pes():
    return get_env().world_size


def team_my_pe(team=NVSHMEM_TEAM_WORLD):
    return get_env().rank


def team_n_pes(team=NVSHMEM_TEAM_WORLD):
    return get_env().world_size


def team_translate_pe(src_team, pe_in_src_team, dest_team):
    return pe_in_src_team


def remote_ptr(local: torch.Tensor, pe: int) -> torch.Tensor:
    return get_env().heap.peer_view(local, pe)


def remote_mc_ptr(team, local):
    """No multicast addresses exist on xGMI."""
    return None


def int_p(dest: torch.Tensor, value: int, pe: int, stream=None):
    env = get_env()
    if env.backend == "host":
        env.heap.peer_view(dest, pe).view(-1)[0] = value
    else:
        set_signal(env.heap.peer_ptr(dest, pe), value, stream, bits=32)


def _put(dest, src, nbytes, pe, stream=None):
    env = get_env()
    d = env.heap.peer_view(dest, pe)
    if env.backend == "host":
        d.view(-1).view(torch.uint8)[:nbytes].copy_(src.reshape(-1).view(torch.uint8)[:nbytes])
    else:
        memcpy_async(d, src, nbytes, stream)


def _get(dest, src, nbytes, pe, stream=None):
    env = get_env()
    s = env.heap.peer_view(src, pe)
    if env.backend == "host":
        dest.view(-1).view(torch.uint8)[:nbytes].copy_(s.reshape(-1).view(torch.uint8)[:nbytes])
    else:
        memcpy_async(dest, s, nbytes, stream)


putmem = putmem_nbi = putmem_block = putmem_nbi_block = putmem_warp = putmem_nbi_warp = _put
getmem = getmem_nbi = getmem_block = getmem_nbi_block = getmem_warp = getmem_nbi_warp = _get
putmem_rma = putmem_rma_nbi = _put


def signal_op(sig_addr: torch.Tensor, signal: int, sig_op: int, pe: int, stream=None):
    notify(sig_addr, pe, signal, "set" if sig_op == NVSHMEM_SIGNAL_SET else "add", "intra_node", stream=stream)


def _put_signal(dest, src, nbytes, sig_addr, signal, sig_op, pe, stream=None):
    _put(dest, src, nbytes, pe, stream)
    signal_op(sig_addr, signal, sig_op, pe, stream)


putmem_signal = putmem_signal_nbi = putmem_signal_block = putmem_signal_nbi_block = _put_signal
putmem_signal_warp = putmem_signal_nbi_warp = putmem_signal_rma = putmem_signal_rma_nbi = _put_signal


def signal_wait_until(sig_addr: torch.Tensor, cmp_: int, cmp_val: int, stream=None, timeout_ms: int = 60000):
    env = get_env()
    bits = sig_addr.element_size() * 8
    if env.backend == "host":
        fn = _lib.host().tdh_wait32 if bits == 32 else _lib.host().tdh_wait64
        ref = ctypes.c_int32(cmp_val) if bits == 32 else ctypes.c_int64(cmp_val)
        if fn(ctypes.c_void_p(sig_addr.data_ptr()), 1, int(cmp_), ref, ctypes.c_int64(timeout_ms)) != 0:
            raise TimeoutError("signal_wait_until timed out")
        return int(sig_addr.view(-1)[0])
    assert cmp_ in (NVSHMEM_CMP_EQ, NVSHMEM_CMP_GE), "stream waits support EQ and GE"
    wait_eq(sig_addr, cmp_val, stream, bits=bits, ge=(cmp_ == NVSHMEM_CMP_GE))
    return cmp_val


def fence(stream=None):
    """Orders earlier puts before later ones: stream order already does (host: program order)."""


def quiet(stream=None):
    env = get_env()
    if env.backend == "hip":
        (stream or torch.cuda.current_stream()).synchronize()


def barrier_all(stream=None):
    from ...utils.compat import nvshmem_barrier_all_on_stream

    nvshmem_barrier_all_on_stream(stream)


barrier_all_block = barrier_all_warp = sync_all = sync_all_block = sync_all_warp = barrier_all


def barrier(team=NVSHMEM_TEAM_WORLD, stream=None):
    barrier_all(stream)


team_sync_block = team_sync_warp = barrier


def broadcast(team, dest: torch.Tensor, src: torch.Tensor, nelems: int, pe_root: int, stream=None):
    """Every rank pulls ``nelems`` elements of ``src`` from ``pe_root`` (call between barriers)."""
    _get(dest, src, nelems * src.element_size(), pe_root, stream)


def fcollect(team, dest: torch.Tensor, src: torch.Tensor, nelems: int, stream=None):
    """All-gather: slot r of ``dest`` <- rank r's ``src`` (call between barriers)."""
    env = get_env()
    nb = nelems * src.element_size()
    flat = dest.view(-1).view(torch.uint8)
    for r in range(env.world_size):
        _get(flat[r * nb:(r + 1) * nb], src, nb, r, stream)


broadcast_block = broadcast_warp = broadcast
fcollect_block = fcollect_warp = fcollect


def broadcastmem(team, dest: torch.Tensor, src: torch.Tensor, nbytes: int, pe_root: int, stream=None):
    """Byte-counted ``broadcast``."""
    _get(dest, src, nbytes, pe_root, stream)


def fcollectmem(team, dest: torch.Tensor, src: torch.Tensor, nbytes: int, stream=None):
    """Byte-counted ``fcollect``."""
    flat, env = dest.view(-1).view(torch.uint8), get_env()
    for r in range(env.world_size):
        _get(flat[r * nbytes:(r + 1) * nbytes], src, nbytes, r, stream)


broadcastmem_block = broadcastmem_warp = broadcastmem
fcollectmem_block = fcollectmem_warp = fcollectmem
barrier_block = barrier_warp = barrier
# the "rma" forms address the remote-memory-access transport of an inter-node SHMEM; inside one xGMI node every
# put is a store through the peer mapping, whatever the scope suffix
putmem_rma_block = putmem_rma_warp = putmem_rma_nbi_block = putmem_rma_nbi_warp = _put
putmem_signal_rma_block = putmem_signal_rma_nbi_block = _put_signal
putmem_signal_rma_warp = putmem_signal_rma_nbi_warp = _put_signal
