"""Task-graph -> single persistent kernel runtime (reference: python/triton_dist/mega_triton_kernel/)."""
from . import core  # noqa: F401
from .core import (Task, TaskDependency, TaskType, check_schedule, register_task, registered_task_types,  # noqa: F401
                   round_robin_scheduler, zig_zag_scheduler)
from .model_builder import ModelBuilder  # noqa: F401
from .qwen3 import Qwen3LayerBuilder, Qwen3Model  # noqa: F401
from .layers import TPAttnBuilder, TPMLPBuilder, prepare_cos_sin_cache  # noqa: F401
from .tasks import ConfigBase, FlashAttnConfig, LinearConfig, TASK_KINDS, get_config_factory  # noqa: F401
from .task_api import (CodeGenerator, DeviceProp, Registry, SchedulingStrategy, TaskBase, TaskBuilderBase, enque_tasks,  # noqa: F401
                       registry)
