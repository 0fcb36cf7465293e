"""Per-op task descriptions of the mega-kernel runtime: the tiling knobs an op takes, how many tile tasks
it cuts into, and which interpreter body runs them.

Reference: python/triton_dist/mega_triton_kernel/tasks/*.py (one ``XxxConfig`` / ``XxxTask`` /
``XxxTaskBuilder`` triple per op type, registered in core/registry.py:38-51) and core/config.py:29
(``ConfigBase``).  There a config parameterises generated Triton source; here the interpreter kernel is
fixed (csrc/ops/mega_kernel.hip), so a config only carries the knobs that kernel honours - the column
block of a linear task, the KV split of an attention task, the chunk of a prefetch task - and
``ModelBuilder.make_*`` accepts it through its ``config`` argument.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Optional, Type

from .core import TaskType


@dataclasses.dataclass
class ConfigBase:
    """Tiling knobs of one op (reference: core/config.py:29)."""

    def to_dict(self) -> dict:
        return dataclasses.asdict(self)


@dataclasses.dataclass
class LinearConfig(ConfigBase):
    """``BLOCK_SIZE_N``: output columns per task (16, 32 or 64 = 1 / 2 / 4 column groups of a 16x16x32 MFMA;
    the four waves of the workgroup split K between them when there are fewer than four groups).
    0 = chosen from N (the largest block that still yields one task per compute unit)."""
    BLOCK_SIZE_N: int = 0
    BLOCK_SIZE_M: int = 16          # fixed: decode-shaped tasks (at most 16 tokens)
    BLOCK_SIZE_K: int = 32          # fixed: one MFMA K-step


@dataclasses.dataclass
class MLPFC1Config(LinearConfig):
    pass


@dataclasses.dataclass
class MLPFC2Config(LinearConfig):
    pass


@dataclasses.dataclass
class QKVProjConfig(LinearConfig):
    pass


@dataclasses.dataclass
class OProjConfig(LinearConfig):
    pass


@dataclasses.dataclass
class FlashAttnConfig(ConfigBase):
    """``NUM_KV_SPLITS``: KV ranges per (batch, head group); 0 = from the queue count and the cache length."""
    NUM_KV_SPLITS: int = 0


@dataclasses.dataclass
class RMSNormConfig(ConfigBase):
    pass


@dataclasses.dataclass
class ElementwiseConfig(ConfigBase):
    pass


@dataclasses.dataclass
class ActivationConfig(ConfigBase):
    pass


@dataclasses.dataclass
class AllReduceConfig(ConfigBase):
    pass


@dataclasses.dataclass
class BarrierConfig(ConfigBase):
    pass


@dataclasses.dataclass
class PrefetchConfig(ConfigBase):
    CHUNK_BYTES: int = 1 << 20


@dataclasses.dataclass
class TaskKind:
    """What the registry knows about one op type: the interpreter body that runs its tiles, its config class
    and the ``ModelBuilder`` method that records it."""
    op_type: str
    task_type: TaskType
    config_cls: Type[ConfigBase]
    builder_method: str
    device_body: str            # name of the device function in csrc/ops/mega_kernel.hip


TASK_KINDS: Dict[str, TaskKind] = {k.op_type: k for k in [
    TaskKind("linear", TaskType.LINEAR, LinearConfig, "make_linear", "task_linear"),
    TaskKind("mlp_fc1", TaskType.LINEAR, MLPFC1Config, "make_fc1", "task_linear (mode 1: silu(gate) * up epilogue)"),
    TaskKind("mlp_fc2", TaskType.LINEAR, MLPFC2Config, "make_fc2", "task_linear"),
    TaskKind("qkv_proj", TaskType.LINEAR, QKVProjConfig, "make_qkv_proj", "task_linear"),
    TaskKind("o_proj", TaskType.LINEAR, OProjConfig, "make_o_proj", "task_linear"),
    TaskKind("rms_norm", TaskType.RMS_NORM, RMSNormConfig, "make_rms_norm", "task_rms_norm"),
    TaskKind("add", TaskType.ADD, ElementwiseConfig, "make_add", "task_add"),
    TaskKind("silu_mul_up", TaskType.SILU_MUL, ActivationConfig, "make_silu_mul_up", "task_silu_mul"),
    TaskKind("qk_norm_rope_update_kvcache", TaskType.QK_NORM_ROPE_KV, ElementwiseConfig,
             "make_qk_norm_rope_update_kvcache", "task_qk_norm_rope_kv"),
    TaskKind("attn_split", TaskType.ATTN_SPLIT, FlashAttnConfig, "make_attn", "task_attn_split"),
    TaskKind("attn_combine", TaskType.ATTN_COMBINE, FlashAttnConfig, "make_attn", "task_attn_combine"),
    TaskKind("allreduce", TaskType.ALLREDUCE, AllReduceConfig, "make_allreduce", "task_allreduce"),
    TaskKind("barrier_all_intra_node", TaskType.BARRIER, BarrierConfig, "make_barrier_all_intra_node",
             "shmem::barrier_all_block"),
    TaskKind("prefetch", TaskType.PREFETCH, PrefetchConfig, "make_prefetch", "task_prefetch"),
]}


def get_config_factory(op_type: str) -> Type[ConfigBase]:
    """Config class of an op type (reference: core/registry.py:63 ``get_config_factory``)."""
    return TASK_KINDS[op_type].config_cls


def default_config(op_type: str) -> ConfigBase:
    return get_config_factory(op_type)()


def linear_block_n(config: Optional[LinearConfig]) -> int:
    bn = int(getattr(config, "BLOCK_SIZE_N", 0) or 0)
    if bn not in (0, 16, 32, 64):
        raise ValueError(f"LinearConfig.BLOCK_SIZE_N must be 16, 32 or 64 (or 0 = automatic), got {bn}")
    return bn
