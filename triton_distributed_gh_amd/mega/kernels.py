"""The task bodies of the mega kernel and their record format, as seen from Python.

Reference: python/triton_dist/mega_triton_kernel/kernels/*.py - Triton device functions, one per task type
(linear.py, mlp_fc1.py, norm.py, activation.py, elementwise.py, flash_attn.py, allreduce.py, barrier.py,
prefetch.py) plus the scoreboard helpers of task_context.py:31-138.  The bodies here are HIP device
functions compiled into the interpreter kernel (csrc/ops/mega_kernel.hip); this module names them and exposes
the two sizes the host needs to lay out queues and the op table.
"""
from __future__ import annotations

from typing import Dict

from .. import _lib
from .core import TASK_WORDS, MegaOpStruct, TaskType
from .tasks import TASK_KINDS

#: task type -> device function that runs one tile of it
TASK_BODIES: Dict[TaskType, str] = {}
for _kind in TASK_KINDS.values():
    TASK_BODIES.setdefault(_kind.task_type, _kind.device_body)

#: layout of one task record (int32 words): the scoreboard protocol of the reference's task_context
TASK_RECORD = ("type", "op", "tile", "ndeps", "dep0_op", "dep0_lo", "dep0_hi", "dep1_op", "dep1_lo", "dep1_hi",
               "dep2_op", "dep2_lo", "dep2_hi", "pad", "pad", "pad")
assert len(TASK_RECORD) == TASK_WORDS


def native_task_words() -> int:
    """Words per task record as compiled into the interpreter (must equal ``core.TASK_WORDS``)."""
    return int(_lib.hip().tdist_mega_task_words())


def native_op_size() -> int:
    """Bytes of one op descriptor as compiled into the interpreter (must equal ``sizeof(MegaOpStruct)``)."""
    return int(_lib.hip().tdist_mega_op_size())


def check_abi() -> None:
    """The Python mirrors and the compiled kernel agree on the record and descriptor sizes."""
    import ctypes

    if native_task_words() != TASK_WORDS or native_op_size() != ctypes.sizeof(MegaOpStruct):
        raise RuntimeError("mega kernel ABI mismatch: rebuild the native library")
