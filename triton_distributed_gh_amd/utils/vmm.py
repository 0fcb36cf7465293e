"""Symmetric arena of any size on the HIP virtual-memory-management API.

A hipIpc mapping stays below 2 GiB on this stack, which caps a symmetric tensor of the hipIpc arenas
(``utils/symm.py``) at one segment.  This arena has no such cap: every rank reserves one contiguous
virtual range per rank of the node, physical memory is committed chunk by chunk (collectively, on
demand), each chunk is exported as a dmabuf file descriptor and mapped by every peer at the same
offset of the range it reserved for the owner.  A range is contiguous, so a tensor may span any
number of chunks and the peer translation stays ``base[peer] + offset``; only committed bytes cost
memory, so the capacity can be the whole 288 GB device.

File descriptors cross process boundaries as SCM_RIGHTS messages on abstract-namespace Unix sockets
(one listening socket per rank; senders connect, send and close, nobody blocks on a peer).

Importing a peer's descriptor needs a driver that keeps the buffer metadata on the dmabuf
(``hsa_amd_vmem_import_shareable_handle``).  Where it does not - the virtualised boxes this package was
developed on answer INCOMPATIBLE_ARGUMENTS to a descriptor of another process, while the same call
succeeds inside one process - ``commit`` raises ``VmmUnsupported`` on EVERY rank, and the heap keeps
to its hipIpc segments.  One process (world size 1, or the exporter itself) always works.

Role in the reference: ``rocshmem_malloc`` tensors of arbitrary size on the rocSHMEM symmetric heap
(reference: shmem/rocshmem_bind/pyrocshmem/src/pyrocshmem.cc:101-113) and NVSHMEM's
``nvshmem_create_tensor`` (python/triton_dist/utils.py:114-136).
"""
from __future__ import annotations

import ctypes
import os
import socket
import struct
import uuid
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import _lib

_U64 = ctypes.c_uint64


class VmmUnsupported(RuntimeError):
    """The driver cannot share hipMemCreate chunks between processes on this machine."""


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class VmmArena:
    """Collective object: construct, ``commit`` and ``close`` on every rank of ``group`` in the same order."""

    def __init__(self, rank: int, world_size: int, group: Optional[dist.ProcessGroup], capacity: int,
                 chunk_bytes: int = 1 << 30, uncached: bool = False):
        lib = _lib.hip()
        self.rank, self.world_size, self.group = rank, world_size, group
        self.device = torch.cuda.current_device()
        self.uncached = 1 if uncached else 0
        g = _U64()
        _lib.check(lib.tdist_vmm_granularity(self.device, self.uncached, ctypes.byref(g)), "hipMemGetAllocationGranularity")
        self.granularity = max(int(g.value), 1 << 21)
        # a chunk stays below 2 GiB: the export / import path shares the driver's dmabuf machinery with hipIpc
        self.chunk = min(_round_up(int(chunk_bytes), self.granularity), _round_up((1 << 31) - (1 << 21), self.granularity)
                         - self.granularity)
        self.capacity = _round_up(int(capacity), self.chunk)
        self.committed = 0
        self._maps: List[tuple] = []          # (va, bytes, handle) of everything mapped by this process
        self.bases: List[int] = []
        for _ in range(world_size):
            p = ctypes.c_void_p()
            _lib.check(lib.tdist_vmm_reserve(ctypes.byref(p), _U64(self.capacity), _U64(self.granularity)),
                       f"hipMemAddressReserve({self.capacity} bytes)")
            self.bases.append(p.value)
        self._sock = None
        if world_size > 1:
            token = self._exchange(uuid.uuid4().hex[:12] if rank == 0 else None)[0]
            self._addr = [f"\0tdist_vmm_{token}_{r}" for r in range(world_size)]
            self._sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            self._sock.bind(self._addr[rank])
            self._sock.listen(max(world_size, 8))
            self._exchange("listening")
        self._closed = False

    def _exchange(self, obj):
        if self.world_size == 1:
            return [obj]
        out = [None] * self.world_size
        dist.all_gather_object(out, obj, group=self.group)
        return out

    # ------------------------------------------------------------------ commit
    def commit(self, upto: int) -> None:
        """Make bytes [0, upto) of every rank's range usable (collective; rounds up to whole chunks)."""
        upto = min(_round_up(int(upto), self.chunk), self.capacity)
        lib = _lib.hip()
        while self.committed < upto:
            off = self.committed
            h = _U64()
            _lib.check(lib.tdist_vmm_create(self.device, self.uncached, _U64(self.chunk), ctypes.byref(h)),
                       f"hipMemCreate({self.chunk} bytes)")
            va = self.bases[self.rank] + off
            _lib.check(lib.tdist_vmm_map(ctypes.c_void_p(va), _U64(self.chunk), h, self.device), "hipMemMap(own chunk)")
            self._maps.append((va, self.chunk, h.value))
            _lib.check(lib.tdist_memset(ctypes.c_void_p(va), 0, ctypes.c_size_t(self.chunk)), "chunk memset")
            if self.world_size > 1:
                fd = ctypes.c_int(-1)
                _lib.check(lib.tdist_vmm_export_fd(h, ctypes.byref(fd)), "hipMemExportToShareableHandle")
                header = struct.pack("<iq", self.rank, off)
                try:
                    for i in range(1, self.world_size):
                        peer = (self.rank + i) % self.world_size
                        with socket.socket(socket.AF_UNIX, socket.SOCK_STREAM) as c:
                            c.connect(self._addr[peer])
                            socket.send_fds(c, [header], [fd.value])
                finally:
                    os.close(fd.value)
                err = None
                for _ in range(self.world_size - 1):
                    conn, _ = self._sock.accept()
                    with conn:
                        msg, fds, _, _ = socket.recv_fds(conn, len(header), 1)
                    src, src_off = struct.unpack("<iq", msg)
                    assert src != self.rank and src_off == off and len(fds) == 1, (src, src_off, off, fds)
                    ph = _U64()
                    try:
                        _lib.check(lib.tdist_vmm_import_fd(fds[0], ctypes.byref(ph)), f"hipMemImportFromShareableHandle(rank {src})")
                        pva = self.bases[src] + off
                        _lib.check(lib.tdist_vmm_map(ctypes.c_void_p(pva), _U64(self.chunk), ph, self.device),
                                   f"hipMemMap(rank {src} chunk)")
                        self._maps.append((pva, self.chunk, ph.value))
                    except _lib.NativeError as exc:      # keep receiving: every sender's message must be drained
                        err = err or str(exc)
                    finally:
                        os.close(fds[0])
                # nobody touches a peer's new chunk before it is mapped everywhere; a failure anywhere fails everywhere
                errs = [e for e in self._exchange(err) if e]
                if errs:
                    raise VmmUnsupported("a peer's chunk could not be imported (the driver must support dmabuf export with "
                                         "buffer metadata across processes): " + errs[0])
            self.committed += self.chunk

    # ------------------------------------------------------------------ teardown
    def close(self, sync_group=None) -> None:
        if self._closed:
            return
        self._closed = True
        lib = _lib.hip()
        try:
            lib.tdist_device_sync()
            if sync_group is not None:
                dist.barrier(group=sync_group)
            for va, nbytes, h in self._maps:
                lib.tdist_vmm_unmap(ctypes.c_void_p(va), _U64(nbytes))
                lib.tdist_vmm_release(_U64(h))
            for b in self.bases:
                lib.tdist_vmm_unreserve(ctypes.c_void_p(b), _U64(self.capacity))
        except Exception:
            pass
        if self._sock is not None:
            self._sock.close()


def default_capacity() -> int:
    """TDIST_SYMM_VMM_BYTES, or 7/8 of the device's memory (virtual address space only: nothing is
    committed until a tensor needs it)."""
    env = os.environ.get("TDIST_SYMM_VMM_BYTES")
    if env:
        return int(env)
    total = torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory
    return total // 8 * 7
