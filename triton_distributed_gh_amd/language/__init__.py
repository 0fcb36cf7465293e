"""Tile-centric distributed primitives: the Python face of csrc/include/tdist/primitives.hpp.

In the reference these names are Triton builtins lowered through the ``distributed`` MLIR dialect
(reference: python/triton_dist/language/__init__.py:25-44, distributed_ops.py:56-111).  Here device
code is hand-written HIP, so kernels call the ``tdist::wait / notify / symm_at ...`` device functions
directly; this module offers the same operations at HOST level - on symmetric tensors, ordered on a
HIP stream (GPU backend) or executed immediately with C++ atomics (host backend) - which is what
protocol prototypes, tutorials and tests need.  Enum values equal the dialect's
(include/TritonDistributed/Dialect/Distributed/IR/DistributedAttrDefs.td:36-53).
"""
from .core import extern_call  # noqa: F401
from .distributed_ops import (COMM_SCOPE, SIGNAL_OP, consume_token, notify, num_ranks, rank, symm_at,  # noqa: F401
                              wait)
from .simt_ops import extract, insert, simt_exec_region  # noqa: F401

__all__ = ["wait", "consume_token", "rank", "num_ranks", "symm_at", "notify", "simt_exec_region", "extract",
           "insert", "extern_call", "SIGNAL_OP", "COMM_SCOPE"]
