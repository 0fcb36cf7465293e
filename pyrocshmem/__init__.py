"""Import-compatible stand-in for the reference's ``pyrocshmem`` extension
(reference: shmem/rocshmem_bind/pyrocshmem/src/pyrocshmem.cc:348-434 and
shmem/rocshmem_bind/python/pyrocshmem/__init__.py:43-108).

The reference binds an IPC-only rocSHMEM build; this framework owns its symmetric heap (hipIpc
arenas exchanged once at start-up, csrc/runtime/runtime_hip.cc + utils/symm.py), so the functions
below resolve to that runtime.  ``rocshmem_get_uniqueid`` / ``rocshmem_init_attr`` keep the
reference's two-step bootstrap shape: the id is opaque and initialisation happens through
``torch.distributed``."""
from __future__ import annotations

import os
from typing import List

import torch

from triton_distributed_gh_amd.utils import distributed as _d
from triton_distributed_gh_amd.utils import get_env, is_initialized
from triton_distributed_gh_amd.utils.compat import nvshmem_barrier_all_on_stream


def rocshmem_get_uniqueid() -> bytes:
    return os.urandom(16)


def rocshmem_init_attr(rank: int, nranks: int, uid: bytes = b"") -> None:
    os.environ.setdefault("RANK", str(rank))
    os.environ.setdefault("WORLD_SIZE", str(nranks))
    _d.initialize_distributed()


def rocshmem_init() -> None:
    _d.initialize_distributed()


def rocshmem_finalize() -> None:
    _d.finalize_distributed()


def rocshmem_my_pe() -> int:
    return get_env().rank


def rocshmem_n_pes() -> int:
    return get_env().world_size


def rocshmem_team_my_pe(team=0) -> int:
    return get_env().rank


def rocshmem_team_n_pes(team=0) -> int:
    return get_env().world_size


def rocshmem_barrier_all() -> None:
    nvshmem_barrier_all_on_stream(None)
    if get_env().backend == "hip":
        torch.cuda.current_stream().synchronize()


def rocshmem_malloc(nbytes: int) -> int:
    return get_env().heap.create_tensor([nbytes], torch.uint8).data_ptr()


def rocshmem_ptr(ptr_or_tensor, pe: int) -> int:
    heap = get_env().heap
    if isinstance(ptr_or_tensor, torch.Tensor):
        return heap.peer_ptr(ptr_or_tensor, pe)
    return heap.translate(int(ptr_or_tensor), pe)


class SymmHeap:
    """Raw view of this rank's data arena (``__cuda_array_interface__`` like the reference's)."""

    def __init__(self):
        heap = get_env().heap
        self.ptr, self.nbytes = heap.base("data", heap.rank), heap.heap_bytes
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False),
                                         "version": 2}


def symm_heap_tensor(tensor: torch.Tensor, peer: int) -> torch.Tensor:
    return get_env().heap.peer_view(tensor, peer)


def rocshmem_create_tensor(shape, dtype) -> torch.Tensor:
    return get_env().heap.create_tensor(list(shape), dtype)


def rocshmem_create_tensor_list_intra_node(shape, dtype) -> List[torch.Tensor]:
    return get_env().heap.create_tensor_list(list(shape), dtype)


hipipc_create_tensor_list = rocshmem_create_tensor_list_intra_node
rocshmem_create_tensor_list = rocshmem_create_tensor_list_intra_node


def rocshmem_get_tensors_from_ipchandle(*_a, **_k):
    raise NotImplementedError("IPC handles are exchanged once, for the whole arena, at start-up "
                              "(SymmetricHeap._init_hip); per-tensor handles do not exist")


__all__ = [n for n in dir() if n.startswith(("rocshmem_", "hipipc_", "symm_", "Symm"))]
_ = is_initialized
