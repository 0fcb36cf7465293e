"""Symmetric heap: one arena per rank, mapped into every rank of the node.

Design (MI355X-first): instead of one IPC handle exchange per tensor
(reference: shmem/rocshmem_bind/pyrocshmem/src/pyrocshmem.cc:198-256 ``hipipc_create_tensor_list``;
python/triton_dist/utils.py:114-136 ``nvshmem_create_tensor(s)``) each rank allocates ONE data
arena and ONE small signal arena at start-up, the handles are exchanged once over
``torch.distributed`` and symmetric tensors are carved at identical offsets on every rank.
``symm_at(ptr, peer)`` on the device is then a base-pointer translation.

One hipIpc mapping cannot reach 2 GiB on this stack (``MAX_IPC_BYTES``), so the data arena GROWS
in equal segments of at most that size, on demand and collectively: when a symmetric tensor does
not fit the segments that exist, every rank allocates one more, the handles are exchanged, and the
tensor is carved from it.  A tensor of these arenas lives inside ONE segment; the device-side
translation checks the segments in turn.

A tensor that is larger than a segment (or arrives when all segments are full) goes to a third kind
of arena, created on first need: ``utils/vmm.py`` - one contiguous virtual range per rank, physical
chunks committed on demand and shared as dmabuf file descriptors - whose capacity defaults to 7/8 of
the device memory (``TDIST_SYMM_VMM_BYTES``).  ``TDIST_HEAP_KIND=vmm`` sends every data tensor there.

Two backends share this file:
* ``hip``  - ``hipMalloc`` / ``hipExtMallocWithFlags`` + ``hipIpcGet/OpenMemHandle``;
* ``host`` - POSIX shared memory (``/dev/shm``) so the full stack runs on a CPU-only machine.
"""
from __future__ import annotations

import ctypes
import math
import os
import uuid
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from .. import _lib

MAX_RANKS = 16
MAX_EXTRA_SEGMENTS = 7          # data arena = 1 + up to 7 growth segments of equal size


# largest allocation that can be shared through hipIpc (see SymmetricHeap.__init__)
MAX_IPC_BYTES = (1 << 31) - (1 << 21)


class SymmCtxStruct(ctypes.Structure):
    """Mirror of ``tdist::SymmCtx`` (csrc/include/tdist/common.hpp); passed to kernels by value."""

    _fields_ = [
        ("rank", ctypes.c_int),
        ("num_ranks", ctypes.c_int),
        ("heap_base", ctypes.c_uint64 * MAX_RANKS),
        ("sig_base", ctypes.c_uint64 * MAX_RANKS),
        ("heap_bytes", ctypes.c_uint64),
        ("sig_bytes", ctypes.c_uint64),
        ("num_extra", ctypes.c_int),
        ("_pad", ctypes.c_int),
        ("extra_base", (ctypes.c_uint64 * MAX_RANKS) * MAX_EXTRA_SEGMENTS),
        ("vmm_bytes", ctypes.c_uint64),
        ("vmm_base", ctypes.c_uint64 * MAX_RANKS),
    ]


class _RawCuda:
    """Minimal ``__cuda_array_interface__`` carrier so torch can wrap a raw device pointer."""

    def __init__(self, ptr: int, nbytes: int, owner=None):
        self.owner = owner
        self.__cuda_array_interface__ = {
            "shape": (nbytes,),
            "typestr": "|u1",
            "data": (ptr, False),
            "version": 2,
        }


def _wrap_device(ptr: int, nbytes: int, owner) -> torch.Tensor:
    return torch.as_tensor(_RawCuda(ptr, nbytes, owner), device=torch.device("cuda", torch.cuda.current_device()))


def _wrap_host(ptr: int, nbytes: int, owner) -> torch.Tensor:
    buf = (ctypes.c_ubyte * nbytes).from_address(ptr)
    t = torch.frombuffer(buf, dtype=torch.uint8)
    t._tdist_owner = owner  # keep the mapping alive
    return t


class _FreeList:
    """First-fit allocator with coalescing; deterministic, so every rank computes equal offsets."""

    def __init__(self, size: int):
        self.size = size
        self.free = [(0, size)]
        self.used = {}

    def alloc(self, nbytes: int, align: int) -> int:
        nbytes = max(int(nbytes), 1)
        for i, (off, sz) in enumerate(self.free):
            start = (off + align - 1) // align * align
            end = start + nbytes
            if end <= off + sz:
                pieces = []
                if start > off:
                    pieces.append((off, start - off))
                if end < off + sz:
                    pieces.append((end, off + sz - end))
                self.free[i : i + 1] = pieces
                self.used[start] = nbytes
                return start
        raise MemoryError(
            f"symmetric arena exhausted: need {nbytes} bytes, arena {self.size}; raise TDIST_SYMM_HEAP_BYTES / "
            "TDIST_SYMM_SIG_BYTES"
        )

    def release(self, off: int) -> None:
        sz = self.used.pop(off)
        self.free.append((off, sz))
        self.free.sort()
        merged = []
        for o, s in self.free:
            if merged and merged[-1][0] + merged[-1][1] == o:
                merged[-1] = (merged[-1][0], merged[-1][1] + s)
            else:
                merged.append((o, s))
        self.free = merged


@dataclass
class SymmTensorInfo:
    arena: str
    offset: int
    nbytes: int


class SymmetricHeap:
    """Collective object: construct it on every rank of ``group`` in the same order."""

    def __init__(
        self,
        rank: int,
        world_size: int,
        group: Optional[dist.ProcessGroup],
        backend: str,
        heap_bytes: Optional[int] = None,
        sig_bytes: Optional[int] = None,
        data_mode: Optional[int] = None,
    ):
        assert world_size <= MAX_RANKS, f"at most {MAX_RANKS} ranks per node"
        self.rank, self.world_size, self.group, self.backend = rank, world_size, group, backend
        # memory mode of the data arena and its growth segments: 0 = coarse-grained hipMalloc,
        # 1 = fine-grained, 2 = uncached (csrc/runtime/runtime_hip.cc: tdist_malloc)
        self.data_mode = int(os.environ.get("TDIST_HEAP_ALLOC_MODE", "0")) if data_mode is None else int(data_mode)
        default_heap = MAX_IPC_BYTES if backend == "hip" else (1 << 26)
        self.heap_bytes = int(heap_bytes or os.environ.get("TDIST_SYMM_HEAP_BYTES", default_heap))
        # total capacity the arena may grow to (segments are added on demand)
        if backend == "hip" and self.heap_bytes > MAX_IPC_BYTES:
            # hipIpcOpenMemHandle never returns for an allocation of 2 GiB or more on this stack
            # (measured: 2^31 - 4 KiB opens in < 1 ms, 2^31 and 3 GiB hang), so a segment stays below
            self.heap_bytes = MAX_IPC_BYTES
        # total capacity the arena may grow to (equal segments are added on demand)
        self.max_heap_bytes = int(os.environ.get("TDIST_SYMM_HEAP_MAX_BYTES", 0)) or \
            (1 + MAX_EXTRA_SEGMENTS) * self.heap_bytes
        self.sig_bytes = int(sig_bytes or os.environ.get("TDIST_SYMM_SIG_BYTES", 1 << 22))
        self._alloc = {"data": _FreeList(self.heap_bytes), "sig": _FreeList(self.sig_bytes)}
        self._bases = {"data": [0] * world_size, "sig": [0] * world_size}
        self._flat = {"data": [None] * world_size, "sig": [None] * world_size}
        self._owned = []
        self._shm_names: List[str] = []
        self._closed = False
        self._vmm, self._vmm_failed = None, ""
        self.heap_kind = os.environ.get("TDIST_HEAP_KIND", "ipc")   # "ipc" | "vmm" (every data tensor on the VMM arena)
        if backend == "hip":
            self._init_hip()
        else:
            self._init_host()
        self._tensor_info = {}

    @classmethod
    def create_virtual(cls, world_size: int, backend: str, heap_bytes: int, sig_bytes: int = 1 << 22):
        """``world_size`` heaps living in THIS process (no IPC): rank r's heap sees every other
        virtual rank's arena at its real address.  Lets one process drive all ranks of a fused op on
        separate streams of a single GPU - the test harness for multi-rank kernels on a 1-GPU box."""
        assert world_size <= MAX_RANKS
        bases = {"data": [], "sig": []}
        owned = []
        for arena, nbytes, mode in (("data", heap_bytes, 0), ("sig", sig_bytes, 1)):
            for _ in range(world_size):
                p = ctypes.c_void_p()
                if backend == "hip":
                    lib = _lib.hip()
                    rc = lib.tdist_malloc(ctypes.byref(p), ctypes.c_size_t(nbytes), mode)
                    if rc != 0 and mode != 0:
                        rc = lib.tdist_malloc(ctypes.byref(p), ctypes.c_size_t(nbytes), 0)
                    _lib.check(rc, "virtual arena allocation")
                    _lib.check(lib.tdist_memset(p, 0, ctypes.c_size_t(nbytes)), "arena memset")
                else:
                    buf = ctypes.create_string_buffer(nbytes)
                    owned.append(buf)
                    p = ctypes.c_void_p(ctypes.addressof(buf))
                bases[arena].append(p.value)
        heaps = []
        vgroup = {"segments": [], "owned": owned}       # growth segments shared by the virtual ranks
        for r in range(world_size):
            h = cls.__new__(cls)
            h._vgroup = vgroup
            h.max_heap_bytes = (1 + MAX_EXTRA_SEGMENTS) * heap_bytes
            h.rank, h.world_size, h.group, h.backend = r, world_size, None, backend
            h.heap_bytes, h.sig_bytes = heap_bytes, sig_bytes
            h._alloc = {"data": _FreeList(heap_bytes), "sig": _FreeList(sig_bytes)}
            h._bases = {a: list(bases[a]) for a in bases}
            wrap = _wrap_device if backend == "hip" else _wrap_host
            h._flat = {a: [wrap(bases[a][p], heap_bytes if a == "data" else sig_bytes, owned) for p in range(world_size)]
                       for a in bases}
            h._owned = [bases[a][r] for a in bases] if backend == "hip" else []
            h._opened, h._shm_names, h._closed, h._tensor_info = [], [], False, {}
            h._virtual = True
            h._vmm, h._vmm_failed, h.heap_kind = None, "", "ipc"
            h.data_mode = 0
            heaps.append(h)
        return heaps

    # ------------------------------------------------------------------ backends
    def _exchange(self, obj):
        if self.world_size == 1:
            return [obj]
        out = [None] * self.world_size
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def _init_hip(self) -> None:
        lib = _lib.hip()
        hsz = lib.tdist_ipc_handle_size()
        sig_mode = int(os.environ.get("TDIST_SIG_ALLOC_MODE", "1"))  # 1 = fine-grained
        data_mode = self.data_mode
        handles = {}
        for arena, nbytes, mode in (("data", self.heap_bytes, data_mode), ("sig", self.sig_bytes, sig_mode)):
            p = ctypes.c_void_p()
            rc = lib.tdist_malloc(ctypes.byref(p), ctypes.c_size_t(nbytes), mode)
            if rc != 0 and mode != 0 and arena == "sig":
                rc = lib.tdist_malloc(ctypes.byref(p), ctypes.c_size_t(nbytes), 0)
            _lib.check(rc, f"symmetric {arena} arena allocation ({nbytes} bytes, mode {mode})")
            _lib.check(lib.tdist_memset(p, 0, ctypes.c_size_t(nbytes)), "arena memset")
            self._owned.append(p.value)
            self._bases[arena][self.rank] = p.value
            if self.world_size > 1:
                h = ctypes.create_string_buffer(hsz)
                _lib.check(lib.tdist_ipc_get_handle(p, h), f"hipIpcGetMemHandle({arena})")
                handles[arena] = bytes(h.raw)
        all_handles = self._exchange((os.getpid(), handles))
        self._opened = []
        for r, (pid, hs) in enumerate(all_handles):
            if r == self.rank:
                continue
            for arena in ("data", "sig"):
                q = ctypes.c_void_p()
                _lib.check(lib.tdist_ipc_open_handle(hs[arena], ctypes.byref(q)), f"hipIpcOpenMemHandle(rank {r}, {arena})")
                self._bases[arena][r] = q.value
                self._opened.append(q.value)
        for arena, nbytes in (("data", self.heap_bytes), ("sig", self.sig_bytes)):
            for r in range(self.world_size):
                self._flat[arena][r] = _wrap_device(self._bases[arena][r], nbytes, self)

    def _init_host(self) -> None:
        lib = _lib.host()
        token = self._token = self._exchange(uuid.uuid4().hex[:12] if self.rank == 0 else None)[0]
        for arena, nbytes in (("data", self.heap_bytes), ("sig", self.sig_bytes)):
            name = f"/tdist_{token}_{arena}_{self.rank}"
            p = ctypes.c_void_p()
            _lib.check_host(lib.tdh_shm_map(name.encode(), ctypes.c_uint64(nbytes), 1, ctypes.byref(p)), "shm create")
            ctypes.memset(p.value, 0, nbytes)
            self._bases[arena][self.rank] = p.value
            self._shm_names.append(name)
        self._exchange("created")  # everyone's arenas exist
        for r in range(self.world_size):
            if r == self.rank:
                continue
            for arena, nbytes in (("data", self.heap_bytes), ("sig", self.sig_bytes)):
                name = f"/tdist_{token}_{arena}_{r}"
                p = ctypes.c_void_p()
                _lib.check_host(lib.tdh_shm_map(name.encode(), ctypes.c_uint64(nbytes), 0, ctypes.byref(p)), "shm open")
                self._bases[arena][r] = p.value
        self._exchange("mapped")
        for name in self._shm_names:  # mappings stay valid after unlink; nothing leaks on a crash
            lib.tdh_shm_unlink(name.encode())
        for arena, nbytes in (("data", self.heap_bytes), ("sig", self.sig_bytes)):
            for r in range(self.world_size):
                self._flat[arena][r] = _wrap_host(self._bases[arena][r], nbytes, self)

    # ------------------------------------------------------------------ allocation
    def _data_arenas(self) -> List[str]:
        """The hipIpc data segments ("data", "data1", ...), in creation order."""
        return [a for a in self._alloc if a.startswith("data")]

    def _arena_size(self, arena: str) -> int:
        if arena == "sig":
            return self.sig_bytes
        if arena == "vmm":
            return self._vmm.capacity
        return self.heap_bytes

    def _carve(self, arena: str, shape: Sequence[int], dtype: torch.dtype, align: int):
        """Returns (arena name actually used, offset, nbytes)."""
        numel = math.prod(shape) if len(shape) else 1
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        if arena != "data":
            return arena, self._alloc[arena].alloc(nbytes, align), nbytes
        can_vmm = self.backend == "hip" and not getattr(self, "_virtual", False)
        if can_vmm and self.heap_kind == "vmm":
            return self._carve_vmm(nbytes, align)
        for name in self._data_arenas():
            try:
                return name, self._alloc[name].alloc(nbytes, align), nbytes
            except MemoryError:
                continue
        n_seg = len(self._data_arenas())
        if nbytes > self.heap_bytes or n_seg > MAX_EXTRA_SEGMENTS or (n_seg + 1) * self.heap_bytes > self.max_heap_bytes:
            if can_vmm:
                return self._carve_vmm(nbytes, align)
            raise MemoryError(
                f"symmetric heap exhausted: need {nbytes} bytes, {n_seg} segment(s) of {self.heap_bytes} bytes in use "
                f"(a tensor must fit one segment; capacity {self.max_heap_bytes}: raise TDIST_SYMM_HEAP_BYTES / "
                "TDIST_SYMM_HEAP_MAX_BYTES)")
        name = self._grow()
        return name, self._alloc[name].alloc(nbytes, align), nbytes

    def _carve_vmm(self, nbytes: int, align: int):
        """Collective: carve from the VMM arena (created on first use), committing chunks as needed."""
        from .vmm import VmmArena, VmmUnsupported, default_capacity

        if self._vmm_failed:
            raise MemoryError(f"symmetric tensor of {nbytes} bytes does not fit a hipIpc segment ({self.heap_bytes} bytes) "
                              f"and the VMM arena is unavailable here: {self._vmm_failed}")
        if self._vmm is None:
            chunk = int(os.environ.get("TDIST_SYMM_VMM_CHUNK_BYTES", 1 << 30))
            self._vmm = VmmArena(self.rank, self.world_size, self.group, default_capacity(), chunk,
                                 uncached=self.data_mode != 0)
            self._alloc["vmm"] = _FreeList(self._vmm.capacity)
            self._bases["vmm"] = list(self._vmm.bases)
            self._flat["vmm"] = [_wrap_device(b, self._vmm.capacity, self) for b in self._vmm.bases]
        off = self._alloc["vmm"].alloc(nbytes, max(align, 256))
        try:
            self._vmm.commit(off + nbytes)
        except VmmUnsupported as exc:                 # raised on every rank together
            self._alloc["vmm"].release(off)
            self._vmm_failed = str(exc)
            if self.heap_kind == "vmm":               # everything was asked to live there: fall back to the segments
                self.heap_kind = "ipc"
                return self._carve("data", [nbytes], torch.uint8, align)
            raise MemoryError(f"symmetric tensor of {nbytes} bytes does not fit a hipIpc segment ({self.heap_bytes} "
                              f"bytes) and the VMM arena is unavailable here: {exc}") from exc
        return "vmm", off, nbytes

    def _grow(self) -> str:
        """Collective: add one data segment on every rank (same order everywhere, because symmetric
        allocations are collective)."""
        k = len(self._data_arenas())
        name = f"data{k}"
        nbytes, w = self.heap_bytes, self.world_size
        bases = [0] * w
        if getattr(self, "_virtual", False):
            segs = self._vgroup["segments"]
            if len(segs) < k:                       # first virtual rank to need it allocates for all
                row = []
                for _ in range(w):
                    if self.backend == "hip":
                        p = ctypes.c_void_p()
                        _lib.check(_lib.hip().tdist_malloc(ctypes.byref(p), ctypes.c_size_t(nbytes), 0), "segment")
                        _lib.check(_lib.hip().tdist_memset(p, 0, ctypes.c_size_t(nbytes)), "segment memset")
                        row.append(p.value)
                    else:
                        buf = ctypes.create_string_buffer(nbytes)
                        self._vgroup["owned"].append(buf)
                        row.append(ctypes.addressof(buf))
                segs.append(row)
            bases = list(segs[k - 1])
            if self.backend == "hip":
                self._owned.append(bases[self.rank])
            wrap = _wrap_device if self.backend == "hip" else _wrap_host
        elif self.backend == "hip":
            lib = _lib.hip()
            p = ctypes.c_void_p()
            _lib.check(lib.tdist_malloc(ctypes.byref(p), ctypes.c_size_t(nbytes), self.data_mode),
                       f"symmetric {name} segment")
            _lib.check(lib.tdist_memset(p, 0, ctypes.c_size_t(nbytes)), "segment memset")
            self._owned.append(p.value)
            bases[self.rank] = p.value
            handle = None
            if w > 1:
                h = ctypes.create_string_buffer(lib.tdist_ipc_handle_size())
                _lib.check(lib.tdist_ipc_get_handle(p, h), f"hipIpcGetMemHandle({name})")
                handle = bytes(h.raw)
            for r, hs in enumerate(self._exchange(handle)):
                if r == self.rank:
                    continue
                q = ctypes.c_void_p()
                _lib.check(lib.tdist_ipc_open_handle(hs, ctypes.byref(q)), f"hipIpcOpenMemHandle(rank {r}, {name})")
                bases[r] = q.value
                self._opened.append(q.value)
            wrap = _wrap_device
        else:
            lib = _lib.host()

            def shm_map(r, create):
                p = ctypes.c_void_p()
                _lib.check_host(lib.tdh_shm_map(f"/tdist_{self._token}_{name}_{r}".encode(), ctypes.c_uint64(nbytes),
                                                1 if create else 0, ctypes.byref(p)), "shm segment")
                return p.value

            bases[self.rank] = shm_map(self.rank, True)
            ctypes.memset(bases[self.rank], 0, nbytes)
            self._exchange("created")               # everyone's segment exists before anyone opens one
            for r in range(w):
                if r != self.rank:
                    bases[r] = shm_map(r, False)
            self._exchange("mapped")
            lib.tdh_shm_unlink(f"/tdist_{self._token}_{name}_{self.rank}".encode())
            wrap = _wrap_host
        self._bases[name] = bases
        self._flat[name] = [wrap(bases[r], nbytes, self) for r in range(w)]
        self._alloc[name] = _FreeList(nbytes)
        return name

    def _view(self, arena: str, r: int, off: int, nbytes: int, shape, dtype) -> torch.Tensor:
        flat = self._flat[arena][r][off : off + nbytes]
        t = flat.view(dtype).view(*shape) if len(shape) else flat.view(dtype).view(())
        return t

    def create_tensor(self, shape, dtype, arena: str = "data", align: int = 256, zero: bool = True) -> torch.Tensor:
        """Collective: returns this rank's tensor; ``peer_views`` gives the other ranks' copies."""
        shape = tuple(int(s) for s in shape)
        arena, off, nbytes = self._carve(arena, shape, dtype, align)
        t = self._view(arena, self.rank, off, max(nbytes, 0), shape, dtype)
        self._tensor_info[t.data_ptr()] = SymmTensorInfo(arena, off, nbytes)
        if zero and nbytes:
            t.zero_()
        return t

    def create_tensor_list(self, shape, dtype, arena: str = "data", align: int = 256, zero: bool = True) -> List[torch.Tensor]:
        """Collective: list over ranks of views of the same symmetric tensor (index ``rank`` is local)."""
        t = self.create_tensor(shape, dtype, arena, align, zero)
        return self.peer_views(t)

    def info(self, t: torch.Tensor) -> SymmTensorInfo:
        base = t.data_ptr()
        for arena in self._alloc:
            b = self._bases[arena][self.rank]
            size = self._arena_size(arena)
            if b <= base < b + size:
                return SymmTensorInfo(arena, base - b, t.numel() * t.element_size())
        raise ValueError("tensor is not inside this rank's symmetric heap")

    def peer_views(self, t: torch.Tensor) -> List[torch.Tensor]:
        inf = self.info(t)
        assert t.is_contiguous(), "symmetric views require a contiguous tensor"
        nbytes = t.numel() * t.element_size()
        out = []
        for r in range(self.world_size):
            if r == self.rank:
                out.append(t)
            else:
                out.append(self._view(inf.arena, r, inf.offset, nbytes, tuple(t.shape), t.dtype))
        return out

    def peer_view(self, t: torch.Tensor, peer: int) -> torch.Tensor:
        return self.peer_views(t)[peer]

    def peer_ptr(self, t: torch.Tensor, peer: int) -> int:
        inf = self.info(t)
        return self._bases[inf.arena][peer] + inf.offset

    def free_tensor(self, t: torch.Tensor) -> None:
        inf = self._tensor_info.pop(t.data_ptr(), None)
        if inf is not None:
            self._alloc[inf.arena].release(inf.offset)

    # ------------------------------------------------------------------ device ctx
    def ctx_struct(self) -> SymmCtxStruct:
        c = SymmCtxStruct()
        c.rank, c.num_ranks = self.rank, self.world_size
        for r in range(self.world_size):
            c.heap_base[r] = self._bases["data"][r]
            c.sig_base[r] = self._bases["sig"][r]
        c.heap_bytes, c.sig_bytes = self.heap_bytes, self.sig_bytes
        extra = [a for a in self._data_arenas() if a != "data"]
        c.num_extra = len(extra)
        for k, name in enumerate(extra):
            for r in range(self.world_size):
                c.extra_base[k][r] = self._bases[name][r]
        if self._vmm is not None:
            c.vmm_bytes = self._vmm.capacity
            for r in range(self.world_size):
                c.vmm_base[r] = self._vmm.bases[r]
        return c

    def translate(self, ptr: int, peer: int) -> int:
        """Host-side twin of the device ``symm_at``: the address of the same symmetric byte on ``peer``."""
        for arena in self._alloc:
            b = self._bases[arena][self.rank]
            size = self._arena_size(arena)
            if b <= ptr < b + size:
                return self._bases[arena][peer] + (ptr - b)
        raise ValueError("pointer is not inside this rank's symmetric heap")

    def base(self, arena: str, r: int) -> int:
        return self._bases[arena][r]

    # ------------------------------------------------------------------ teardown
    def close(self, sync_group=None) -> None:
        """Unmap the peers' arenas and free this rank's.  ``sync_group``: process group to barrier on
        between the two steps, for a collective close while every rank is alive (nobody frees an arena
        a peer still has mapped)."""
        if self._closed:
            return
        self._closed = True
        self._flat = {"data": [], "sig": []}
        if self._vmm is not None:
            self._vmm.close(sync_group)
        if getattr(self, "_virtual", False) and self.backend == "host":
            return
        if self.backend == "hip":
            lib = _lib.hip()
            try:
                lib.tdist_device_sync()
                for p in getattr(self, "_opened", []):
                    lib.tdist_ipc_close_handle(ctypes.c_void_p(p))
                if sync_group is not None:
                    dist.barrier(group=sync_group)
                for p in self._owned:
                    lib.tdist_free(ctypes.c_void_p(p))
            except Exception:
                pass
        else:
            lib = _lib.host()
            for arena in list(self._bases):
                nbytes = self.sig_bytes if arena == "sig" else self.heap_bytes
                for r in range(self.world_size):
                    b = self._bases[arena][r]
                    if b:
                        lib.tdh_shm_unmap(ctypes.c_void_p(b), ctypes.c_uint64(nbytes))
