"""Task graph, registry and schedulers of the mega-kernel runtime.

Reference: python/triton_dist/mega_triton_kernel/core/ - ``task_base.py`` (TaskBase, TaskDependency,
``encoding()`` int tuple :198-215), ``registry.py`` (``register_task`` :74), ``scheduler.py``
(round-robin / zig-zag work queues :40-95), ``builder.py``.  The reference encodes tensor
descriptors inside every task record and generates kernel source per model; here a task record is
16 int32 words ``[type, op, tile, ndeps, 3 x (dep op, tile lo, tile hi)]`` and the operands live in a
per-op descriptor table consumed by one fixed interpreter kernel (csrc/ops/mega_kernel.hip).
"""
from __future__ import annotations

import ctypes
import dataclasses
import enum
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

TASK_WORDS = 16
MAX_DEPS = 3
MAX_NUM_TENSOR_DIMS = 4


class TaskType(enum.IntEnum):
    LINEAR = 1
    RMS_NORM = 2
    ADD = 3
    QK_NORM_ROPE_KV = 4
    ATTN_SPLIT = 5
    ATTN_COMBINE = 6
    ALLREDUCE = 7
    BARRIER = 8
    PREFETCH = 9
    SILU_MUL = 10


class MegaOpStruct(ctypes.Structure):
    """Mirror of ``MegaOp`` in csrc/ops/mega_kernel.hip."""

    _fields_ = [("p", ctypes.c_uint64 * 10), ("i", ctypes.c_int32 * 16), ("f", ctypes.c_float * 4)]


@dataclasses.dataclass
class TaskDependency:
    """Tiles [lo, hi) of op ``op`` must be complete."""
    op: int
    lo: int
    hi: int


@dataclasses.dataclass
class Task:
    type: TaskType
    op: int
    tile: int
    deps: List[TaskDependency]

    def encoding(self) -> Tuple[int, ...]:
        assert len(self.deps) <= MAX_DEPS, "a task waits on at most 3 producer ranges"
        words = [int(self.type), self.op, self.tile, len(self.deps)]
        for d in self.deps:
            assert d.hi > d.lo >= 0
            words += [d.op, d.lo, d.hi]
        return tuple(words + [0] * (TASK_WORDS - len(words)))


@dataclasses.dataclass
class OpRecord:
    """One node of the graph: a task type, its operand descriptor and its tile count."""
    op_id: int
    type: TaskType
    name: str
    num_tiles: int
    ptrs: List[int]
    ints: List[int]
    floats: List[float]
    keep: list                       # tensors kept alive
    writes: List[int] = dataclasses.field(default_factory=list)   # data_ptr of outputs (dependency tracking)
    reads: List[int] = dataclasses.field(default_factory=list)

    def pack(self) -> MegaOpStruct:
        s = MegaOpStruct()
        for k, v in enumerate(self.ptrs):
            s.p[k] = int(v)
        for k, v in enumerate(self.ints):
            s.i[k] = int(v)
        for k, v in enumerate(self.floats):
            s.f[k] = float(v)
        return s


# ---- registry (op_type -> tiling / dependency rule), cf. registry.register_task -------------------------
_TASK_BUILDERS: Dict[str, Callable] = {}


def register_task(op_type: str):
    def deco(fn):
        _TASK_BUILDERS[op_type] = fn
        return fn

    return deco


def registered_task_types() -> List[str]:
    return sorted(_TASK_BUILDERS)


def get_task_builder(op_type: str) -> Callable:
    if op_type not in _TASK_BUILDERS:
        raise KeyError(f"no task builder registered for {op_type!r}; known: {registered_task_types()}")
    return _TASK_BUILDERS[op_type]


# ---- schedulers ---------------------------------------------------------------------------------
def round_robin_scheduler(tasks: Sequence[Task], num_queues: int) -> List[List[Task]]:
    """Task i -> queue i % num_queues (tasks are in topological order)."""
    qs: List[List[Task]] = [[] for _ in range(num_queues)]
    for i, t in enumerate(tasks):
        qs[i % num_queues].append(t)
    return qs


def zig_zag_scheduler(tasks: Sequence[Task], num_queues: int) -> List[List[Task]]:
    """Like round robin, but every other sweep runs backwards so the same queue does not always get
    the first (or last) tile of an op - evens out tail effects of short ops."""
    qs: List[List[Task]] = [[] for _ in range(num_queues)]
    for i, t in enumerate(tasks):
        sweep, pos = divmod(i, num_queues)
        qs[pos if sweep % 2 == 0 else num_queues - 1 - pos].append(t)
    return qs


SCHEDULERS = {"round_robin": round_robin_scheduler, "zig_zag": zig_zag_scheduler}


def check_schedule(queues: Sequence[Sequence[Task]]) -> None:
    """Deadlock check by simulation: repeatedly retire any queue head whose dependencies are
    complete; every task must retire."""
    done = set()
    heads = [0] * len(queues)
    remaining = sum(len(q) for q in queues)
    progressed = True
    while remaining and progressed:
        progressed = False
        for qi, q in enumerate(queues):
            while heads[qi] < len(q):
                t = q[heads[qi]]
                if all((d.op, tile) in done for d in t.deps for tile in range(d.lo, d.hi)):
                    done.add((t.op, t.tile))
                    heads[qi] += 1
                    remaining -= 1
                    progressed = True
                else:
                    break
    if remaining:
        raise RuntimeError(f"schedule deadlocks: {remaining} tasks can never start")


def encode_queues(queues: Sequence[Sequence[Task]], device) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """-> (int32 [Q, max_tasks, TASK_WORDS], int32 [Q] lengths, max_tasks)."""
    max_tasks = max(1, max(len(q) for q in queues))
    buf = torch.full((len(queues), max_tasks, TASK_WORDS), -1, dtype=torch.int32)
    for qi, q in enumerate(queues):
        if q:
            buf[qi, : len(q)] = torch.tensor([t.encoding() for t in q], dtype=torch.int32)
    lens = torch.tensor([len(q) for q in queues], dtype=torch.int32)
    return buf.to(device), lens.to(device), max_tasks
