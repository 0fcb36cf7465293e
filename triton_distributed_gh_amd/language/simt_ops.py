"""SIMT region helpers (reference: python/triton_dist/language/simt_ops.py).

Triton programs are tile-level, so the reference needs a ``simt`` dialect to run per-thread code
inside a tile program.  HIP kernels ARE per-thread code; nothing has to be lowered.  The names are
kept so reference-style host prototypes import cleanly: the region is a no-op context manager and
``extract`` / ``insert`` are scalar element access on tensors."""
from __future__ import annotations

import contextlib

import torch


@contextlib.contextmanager
def simt_exec_region(*_args, **_kwargs):
    yield


def extract(tensor: torch.Tensor, indices):
    return tensor[tuple(indices)] if isinstance(indices, (list, tuple)) else tensor[indices]


def insert(tensor: torch.Tensor, scalar, indices):
    tensor[tuple(indices) if isinstance(indices, (list, tuple)) else indices] = scalar
    return tensor
