"""Per-op task classes, task builders and the registry object of the mega-kernel runtime.

Reference: python/triton_dist/mega_triton_kernel/core/task_base.py:36-220 (``CodeGenKey``, ``TaskIDManager``,
``TaskBase``, ``DeviceProp``), core/builder.py:33-64 (``TaskBuilderBase``), core/registry.py:30-74
(``Registry``), core/scheduler.py:30-95 (``SchedulingStrategy``, ``enque_tasks``) and the
``XxxTask`` / ``XxxTaskBuilder`` / ``xxx_config_factory`` / ``codegen_xxx`` quadruple of every tasks/*.py.

There a task carries its tensor descriptors in the queue record and ``codegen_xxx`` emits Triton source for
it.  Here the kernel is one fixed interpreter: a task is the 16-word record of ``core.Task``, operands live in
the op table, and "code generation" is the choice of the device function that runs the tile - so

* ``XxxTask`` describes ONE op's tile range the reference's way (layer, task id, tile range, config, io
  tensors) and lowers to ``core.Task`` records with ``lower()``;
* ``XxxTaskBuilder.build_tasks`` records the op on a ``ModelBuilder`` (the same ``make_*`` call the model
  code uses) and returns one task object per tile;
* ``codegen_xxx(task)`` returns the name of the interpreter body in csrc/ops/mega_kernel.hip.
"""
from __future__ import annotations

import dataclasses
import enum
from typing import Any, Callable, ClassVar, Dict, List, Optional, Sequence, Tuple, Type

import torch

from . import core
from .core import SCHEDULERS, Task, TaskDependency, encode_queues
from .tasks import TASK_KINDS, ConfigBase

UNUSED_KEY = -1


@dataclasses.dataclass(frozen=True)
class CodeGenKey:
    """Identifies the device code of a task: the task type alone, or (type, layer, task) when a layer has its
    own specialisation.  The fixed interpreter only ever keys on the type."""
    task_type: int
    layer_id: int = UNUSED_KEY
    task_id: int = UNUSED_KEY

    def __post_init__(self):
        if (self.layer_id == UNUSED_KEY) != (self.task_id == UNUSED_KEY) and self.task_type != UNUSED_KEY:
            raise ValueError(f"layer id and task id are given together or not at all: {self}")

    def only_use_task_type(self) -> bool:
        return self.layer_id == UNUSED_KEY and self.task_id == UNUSED_KEY


class TaskIDManager:
    """Task ids are dense per layer; type ids are the interpreter's ``TaskType`` values."""
    _next_in_layer: Dict[int, int] = {}

    @classmethod
    def get_task_type_id(cls, task_cls: Type["TaskBase"]) -> int:
        return int(TASK_KINDS[task_cls.OP_TYPE].task_type)

    @classmethod
    def get_task_id(cls, layer_id: int) -> int:
        n = cls._next_in_layer.get(layer_id, 0)
        cls._next_in_layer[layer_id] = n + 1
        return n

    @classmethod
    def reset_task_ids(cls) -> None:
        cls._next_in_layer.clear()

    reset_all_ids = reset_task_ids


@dataclasses.dataclass
class DeviceProp:
    NUM_SMS: int            # persistent workgroups (= work queues); one per compute unit by default


@dataclasses.dataclass
class OutputTilingDesc:
    tile_sizes: Optional[Tuple[int, ...]]


@dataclasses.dataclass
class InputDependencyDesc:
    """Which part of ``input`` a task reads: all of it, or the box ``start_indices`` + ``data_sizes``."""
    input: torch.Tensor
    require_full: bool = True
    start_indices: Tuple[int, ...] = ()
    data_sizes: Tuple[int, ...] = ()

    def __post_init__(self):
        if self.require_full:
            self.start_indices, self.data_sizes = (0, ) * self.input.dim(), tuple(self.input.shape)


@dataclasses.dataclass
class TaskBase:
    layer_id: int
    task_id: int
    tile_id_or_start: int
    num_tiles: int
    config: Optional[ConfigBase]
    dependency: Optional[TaskDependency]
    io_tensors: List[List[torch.Tensor]]
    extra_params: Dict[str, Any] = dataclasses.field(default_factory=dict)
    inputs_dep: Dict[Any, InputDependencyDesc] = dataclasses.field(default_factory=dict)
    outs_tile_mapping: Dict[Any, OutputTilingDesc] = dataclasses.field(default_factory=dict)
    op_id: int = -1             # row of the op table (set by the builder that recorded the op)

    OP_TYPE: ClassVar[str] = ""

    @classmethod
    def get_task_type_id(cls) -> int:
        return TaskIDManager.get_task_type_id(cls)

    @classmethod
    def get_codegen_key(cls, layer_id: int = UNUSED_KEY, task_id: int = UNUSED_KEY) -> CodeGenKey:
        return CodeGenKey(cls.get_task_type_id(), layer_id, task_id)

    def lower(self) -> Task:
        """The interpreter's record of this tile."""
        assert self.op_id >= 0, "the op of this task has not been recorded on a ModelBuilder"
        deps = [self.dependency] if self.dependency is not None and self.dependency.hi > self.dependency.lo else []
        return Task(TASK_KINDS[self.OP_TYPE].task_type, self.op_id, self.tile_id_or_start, deps)

    def encoding(self) -> Tuple[int, ...]:
        return self.lower().encoding()


class TaskBuilderBase:
    """``build_tasks`` records the op on ``builder`` (a scratch ``ModelBuilder`` when none is given) through the
    registered ``make_*`` method, called as ``make(*inputs, *outputs, **extra_params)``."""
    TASK_CLS: ClassVar[Type[TaskBase]] = TaskBase
    _create_config: ClassVar[Optional[Callable]] = None

    @classmethod
    def get_task_id(cls, layer_id: int) -> int:
        return TaskIDManager.get_task_id(layer_id)

    @classmethod
    def create_config(cls, **kwargs) -> ConfigBase:
        if cls._create_config is None:
            raise RuntimeError("config factory not set: the builder is not registered")
        return cls._create_config(**kwargs)

    @classmethod
    def get_problem_size(cls, io_tensors, extra_params):
        return tuple(tuple(t.shape) for t in io_tensors[0] + io_tensors[1])

    @classmethod
    def build_tasks(cls, device_prop: DeviceProp, layer_id: int, dependency: Optional[TaskDependency],
                    io_tensors: List[List[torch.Tensor]], extra_params: Optional[Dict[str, Any]] = None,
                    builder=None) -> List[TaskBase]:
        from .model_builder import ModelBuilder

        extra = dict(extra_params or {})
        kind = TASK_KINDS[cls.TASK_CLS.OP_TYPE]
        b = builder if builder is not None else ModelBuilder(num_queues=device_prop.NUM_SMS)
        config = extra.pop("config", None)
        if config is not None and "config" in getattr(b, kind.builder_method).__code__.co_varnames:
            extra["config"] = config
        made = getattr(b, kind.builder_method)(*io_tensors[0], *io_tensors[1], **extra)
        ops = [op for op in (made if isinstance(made, tuple) else (made, )) if op.type == kind.task_type]
        assert len(ops) == 1
        op, task_id = ops[0], cls.get_task_id(layer_id)
        return [cls.TASK_CLS(layer_id, task_id, tile, op.num_tiles, config, dependency, io_tensors, dict(extra_params or {}),
                             op_id=op.op_id) for tile in range(op.num_tiles)]


class Registry:
    """op type -> (task class, builder class, config factory, device body)."""

    def __init__(self):
        self._builders: Dict[Type[TaskBase], Type[TaskBuilderBase]] = {}
        self._config_factories: Dict[Type[TaskBase], Callable] = {}
        self._codegens: Dict[Type[TaskBase], Callable] = {}
        self._op_mapping: Dict[str, Type[TaskBase]] = {}

    def register_task(self, op_type: str, task_cls: Type[TaskBase], config_factory: Callable, codegen_func: Callable):
        def deco(builder_cls: Type[TaskBuilderBase]):
            self._builders[task_cls], self._config_factories[task_cls] = builder_cls, config_factory
            self._codegens[task_cls], self._op_mapping[op_type] = codegen_func, task_cls
            builder_cls._create_config = staticmethod(config_factory)
            builder_cls.TASK_CLS = task_cls
            return builder_cls

        return deco

    def _get(self, table, key, what):
        if key not in table:
            raise ValueError(f"no {what} registered for {getattr(key, '__name__', key)!r}")
        return table[key]

    def get_op_mapping(self, op_type: str) -> Type[TaskBase]:
        return self._get(self._op_mapping, op_type, "task class")

    def get_builder(self, task_cls) -> Type[TaskBuilderBase]:
        return self._get(self._builders, task_cls, "builder")

    def get_config_factory(self, task_cls) -> Callable:
        return self._get(self._config_factories, task_cls, "config factory")

    def get_codegen(self, task_cls) -> Callable:
        return self._get(self._codegens, task_cls, "device body")

    def op_types(self) -> List[str]:
        return sorted(self._op_mapping)


registry = Registry()

# ---- the fourteen built-in op types ---------------------------------------------------------------------
_CLASS_STEM = {"linear": "Linear", "mlp_fc1": "MLPFC1", "mlp_fc2": "MLPFC2", "qkv_proj": "QKVProj", "o_proj": "OProj",
               "rms_norm": "RMSNorm", "add": "Add", "silu_mul_up": "SiLUMulUp",
               "qk_norm_rope_update_kvcache": "QKNormRopeUpdateKVCache", "attn_split": "AttnSplit",
               "attn_combine": "AttnCombine", "allreduce": "AllReduce", "barrier_all_intra_node": "BarrierAllIntraNode",
               "prefetch": "Prefetch"}


def _codegen_for(op_type: str) -> Callable:
    def codegen(task: Optional[TaskBase] = None) -> str:
        return TASK_KINDS[op_type].device_body

    codegen.__name__ = f"codegen_{op_type}"
    codegen.__doc__ = f"Device function of csrc/ops/mega_kernel.hip that runs one ``{op_type}`` tile."
    return codegen


def _factory_for(op_type: str) -> Callable:
    def factory(**kwargs) -> ConfigBase:
        cls = TASK_KINDS[op_type].config_cls
        known = {f.name for f in dataclasses.fields(cls)}
        return cls(**{k: v for k, v in kwargs.items() if k in known})

    factory.__name__ = f"{op_type}_config_factory"
    return factory


for _op, _stem in _CLASS_STEM.items():
    _task = dataclasses.dataclass(type(f"{_stem}Task", (TaskBase, ), {"OP_TYPE": _op, "__module__": __name__}))
    _cg, _cf = _codegen_for(_op), _factory_for(_op)
    _bld = registry.register_task(_op, _task, _cf, _cg)(type(f"{_stem}TaskBuilder", (TaskBuilderBase, ), {"__module__": __name__}))
    globals().update({f"{_stem}Task": _task, f"{_stem}TaskBuilder": _bld, f"codegen_{_op}": _cg, f"{_op}_config_factory": _cf})

ElementwiseTask = globals()["AddTask"]
elementwise_config_factory = globals()["add_config_factory"]
attn_config_factory = globals()["attn_split_config_factory"]
codegen_silu_mul_up_fc1 = globals()["codegen_mlp_fc1"]
# config classes under the reference's names
from .tasks import ActivationConfig as SiLUMulUpConfig, BarrierConfig as BarrierAllIntraNodeConfig  # noqa: E402
from .tasks import ElementwiseConfig as QKNormRopeUpdateKVCacheConfig, FlashAttnConfig as AttnConfig  # noqa: E402,F401


# ---- scheduling front end (reference: core/scheduler.py:30-95) ------------------------------------------------
class SchedulingStrategy(enum.Enum):
    ROUND_ROBIN = "round_robin"
    ZIG_ZAG = "zig_zag"

    def __eq__(self, other):
        return self.value == other if isinstance(other, str) else super().__eq__(other)

    __hash__ = enum.Enum.__hash__


def work_queue_list_to_device_tensor(queues: Sequence[Sequence[Task]], device=None):
    """-> (int32 [Q, max_tasks, 16] records, int32 [Q] lengths) on ``device`` (default: the current GPU)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    buf, lens, _ = encode_queues(queues, device)
    return buf, lens


def enque_tasks(num_sms: int, megakernel_tasks: Sequence, strategy=SchedulingStrategy.ROUND_ROBIN, device=None):
    """Distribute tasks (``core.Task`` or ``TaskBase`` objects, in topological order) over ``num_sms`` queues."""
    name = strategy.value if isinstance(strategy, SchedulingStrategy) else str(strategy)
    if name not in SCHEDULERS:
        raise NotImplementedError(f"unknown scheduling strategy {strategy}")
    tasks = [t.lower() if isinstance(t, TaskBase) else t for t in megakernel_tasks]
    queues = SCHEDULERS[name](tasks, num_sms)
    core.check_schedule(queues)
    return work_queue_list_to_device_tensor(queues, device)


# ---- host views of what the interpreter reads (reference: kernels/task_context.py:31-138) ----------------------
@dataclasses.dataclass
class TensorDesc:
    """Address and padded shape of an operand (the reference packs this into every task record; here the
    operand addresses are in the op table and this is a debugging view)."""
    data_ptr: int
    shape: Tuple[int, ...]

    @classmethod
    def of(cls, t: torch.Tensor) -> "TensorDesc":
        return cls(t.data_ptr(), tuple(t.shape) + (1, ) * (core.MAX_NUM_TENSOR_DIMS - t.dim()))


@dataclasses.dataclass
class TaskBaseInfo:
    """Decoded task record."""
    task_type: int
    op: int
    tile: int
    deps: List[TaskDependency]

    @classmethod
    def decode(cls, words: Sequence[int]) -> "TaskBaseInfo":
        w = [int(x) for x in words]
        return cls(w[0], w[1], w[2], [TaskDependency(*w[4 + 3 * i:7 + 3 * i]) for i in range(w[3])])


class Scoreboard:
    """Host view of the completion table: word ``op * max_tiles + tile`` holds the number of the last launch
    that finished that tile."""

    def __init__(self, table: torch.Tensor, max_tiles: int):
        self.table, self.max_tiles = table, max_tiles

    def done(self, op: int, tile: int, launch: int) -> bool:
        return int(self.table[op * self.max_tiles + tile]) >= launch

    def pending(self, op: int, num_tiles: int, launch: int) -> List[int]:
        row = self.table[op * self.max_tiles: op * self.max_tiles + num_tiles].cpu()
        return [i for i, v in enumerate(row.tolist()) if v < launch]


# ---- argument checks of the reference's ModelBuilder (models/model_builder.py:44-80) --------------------------
def check_tensor_shape(t: torch.Tensor, shape) -> None:
    assert tuple(t.shape) == tuple(shape), f"shape {tuple(t.shape)} != {tuple(shape)}"


def check_tensor_dim(t: torch.Tensor, dim: int) -> None:
    assert t.dim() == dim, f"{t.dim()} dims != {dim}"


def check_tensor_dtype(t: torch.Tensor, dtype: torch.dtype) -> None:
    assert t.dtype == dtype, f"{t.dtype} != {dtype}"


def check_contiguous(t: torch.Tensor) -> None:
    assert t.is_contiguous(), "tensor must be contiguous"


def check_alignment(t: torch.Tensor, nbytes: int = 16) -> None:
    assert t.data_ptr() % nbytes == 0, f"tensor must be {nbytes}-byte aligned"


def is_multicast_ptr(t: torch.Tensor) -> bool:
    """xGMI has no multicast address space: never."""
    return False


def cdiv(a: int, b: int) -> int:
    return (a + b - 1) // b


def make_mega_kernel_src() -> str:
    """Source text of the interpreter kernel (the reference generates this per model; here it is one fixed file)."""
    from pathlib import Path

    return (Path(__file__).resolve().parents[1] / "csrc" / "ops" / "mega_kernel.hip").read_text()


class CodeGenerator:
    """The reference emits one ``if task_type == ...`` branch per registered codegen function; the fixed
    interpreter's dispatch is the same table, read from the registry."""

    def __init__(self, reg: Registry = registry):
        self.registry = reg

    def dispatch_table(self) -> Dict[int, str]:
        return {c.get_task_type_id(): self.registry.get_codegen(c)() for c in
                (self.registry.get_op_mapping(o) for o in self.registry.op_types())}

    def generate_code(self, tasks=None) -> str:
        return make_mega_kernel_src()
