"""Import-compatible facade: the reference's ``triton_dist`` module tree mapped onto the
MI355X-native implementation in ``triton_distributed_gh_amd`` (reference: python/triton_dist/).

``import triton_dist.kernels.amd``, ``from triton_dist.utils import initialize_distributed``,
``from triton_dist.layers.nvidia import TP_MLP`` ... resolve through the alias table below; every
name lands on a HIP / C++-backed implementation - there is no Triton JIT underneath.  A module
listed with several targets is a synthetic namespace that merges their public names (first wins).
"""
import importlib
import importlib.abc
import importlib.util
import sys
import types

import triton_distributed_gh_amd as _impl

__version__ = _impl.__version__
_P = "triton_distributed_gh_amd."

# facade module -> implementation module(s)
ALIASES = {
    "utils": ["utils.compat"],
    "autotuner": ["autotuner"],
    "language": ["language"],
    "language.distributed_ops": ["language.distributed_ops"],
    "language.simt_ops": ["language.simt_ops"],
    "language.core": ["language.core"],
    "language.extra": ["language.extra"],
    "language.extra.libshmem_device": ["language.extra.libshmem_device"],
    "kernels": [],
    "kernels.allreduce": ["ops.allreduce"],
    "kernels.common_ops": ["ops.common_ops"],
    "kernels.amd": ["ops.ag_gemm", "ops.gemm_rs", "ops.common_ops", "ops.compat_names"],
    "kernels.amd.all_gather_gemm": ["ops.ag_gemm", "ops.compat_names"],
    "kernels.amd.gemm_reduce_scatter": ["ops.gemm_rs", "ops.compat_names"],
    "kernels.amd.common_ops": ["ops.common_ops"],
    "kernels.nvidia": ["ops.compat_nvidia", "ops.allgather", "ops.moe", "ops.flash_decode", "ops.all_to_all",
                       "ops.sp_ag_attention", "ops.gemm_ar", "ops.allreduce", "ops.reduce_scatter", "ops.ep_a2a",
                       "ops.p2p", "ops.common_ops", "ops.compat_names"],
    "kernels.nvidia.allgather_gemm": ["ops.compat_nvidia", "ops.ag_gemm", "ops.compat_names"],
    "kernels.nvidia.gemm_reduce_scatter": ["ops.compat_nvidia", "ops.gemm_rs", "ops.compat_names"],
    "kernels.nvidia.allgather": ["ops.allgather", "ops.compat_nvidia", "ops.compat_names"],
    "kernels.nvidia.low_latency_allgather": ["ops.allgather", "ops.compat_names"],
    "kernels.nvidia.allreduce": ["ops.allreduce", "ops.compat_names"],
    "kernels.nvidia.gemm_allreduce": ["ops.gemm_ar", "ops.compat_names"],
    "kernels.nvidia.reduce_scatter": ["ops.reduce_scatter", "ops.compat_names"],
    "kernels.nvidia.allgather_group_gemm": ["ops.moe", "ops.compat_names"],
    "kernels.nvidia.moe_reduce_rs": ["ops.moe", "ops.compat_names"],
    "kernels.nvidia.moe_utils": ["ops.moe", "ops.compat_names"],
    "kernels.nvidia.threadblock_swizzle_ag_moe": ["ops.threadblock_swizzle", "ops.moe"],
    "kernels.nvidia.low_latency_all_to_all": ["ops.all_to_all", "ops.compat_names"],
    "kernels.nvidia.ep_a2a": ["ops.ep_a2a", "ops.compat_names"],
    "kernels.nvidia.flash_decode": ["ops.flash_decode", "tools.aot_kernels", "ops.compat_names"],
    "kernels.nvidia.sp_ag_attention_intra_node": ["ops.sp_ag_attention", "ops.compat_names"],
    "kernels.nvidia.sp_ag_attention_inter_node": ["ops.sp_ag_attention", "ops.compat_names"],
    "kernels.nvidia.p2p": ["ops.p2p"],
    "kernels.nvidia.common_ops": ["ops.common_ops"],
    "kernels.nvidia.comm_perf_model": ["utils.perf_model", "ops.compat_names"],
    "kernels.nvidia.gemm_perf_model": ["utils.perf_model", "ops.compat_names"],
    "layers": ["layers"],
    "layers.amd": ["layers"],
    "layers.amd.tp_mlp": ["layers.tp_mlp"],
    "layers.amd.tp_attn": ["layers.tp_attn"],
    "layers.nvidia": ["layers"],
    "layers.nvidia.tp_mlp": ["layers.tp_mlp"],
    "layers.nvidia.tp_attn": ["layers.tp_attn"],
    "layers.nvidia.tp_moe": ["layers.tp_moe"],
    "layers.nvidia.ep_a2a_layer": ["ops.ep_a2a"],
    "layers.nvidia.gemm_allreduce_layer": ["layers.gemm_allreduce_layer"],
    "layers.nvidia.low_latency_allgather_layer": ["layers.low_latency_allgather_layer"],
    "layers.nvidia.p2p": ["layers.p2p"],
    "layers.nvidia.sp_flash_decode_layer": ["layers.sp_flash_decode_layer"],
    "models": ["models"],
    "models.config": ["models.config"],
    "models.dense": ["models.dense"],
    "models.qwen_moe": ["models.qwen_moe"],
    "models.kv_cache": ["models.kv_cache"],
    "models.utils": ["models.utils"],
    "models.engine": ["models.engine"],
    "tools": ["tools"],
    "tools.compile_aot": ["tools.compile_aot"],
    "tools.runtime": ["tools.runtime"],
    "mega_triton_kernel": ["mega"],
    "mega_triton_kernel.core": ["mega.core", "mega.task_api"],
    "mega_triton_kernel.core.task_base": ["mega.core", "mega.task_api"],
    "mega_triton_kernel.core.scheduler": ["mega.core", "mega.task_api"],
    "mega_triton_kernel.core.registry": ["mega.core", "mega.task_api"],
    "mega_triton_kernel.core.builder": ["mega.core", "mega.task_api"],
    "mega_triton_kernel.core.config": ["mega.tasks"],
    "mega_triton_kernel.core.code_generator": ["mega.kernels", "mega.task_api"],
    "mega_triton_kernel.tasks": ["mega.tasks", "mega.task_api"],
    "mega_triton_kernel.tasks.linear": ["mega.tasks", "mega.task_api"],
    "mega_triton_kernel.tasks.norm": ["mega.tasks", "mega.task_api"],
    "mega_triton_kernel.tasks.attn": ["mega.tasks", "mega.task_api"],
    "mega_triton_kernel.tasks.activation": ["mega.tasks", "mega.task_api"],
    "mega_triton_kernel.tasks.elementwise": ["mega.tasks", "mega.task_api"],
    "mega_triton_kernel.tasks.allreduce": ["mega.tasks", "mega.task_api"],
    "mega_triton_kernel.tasks.barrier": ["mega.tasks", "mega.task_api"],
    "mega_triton_kernel.tasks.prefetch": ["mega.tasks", "mega.task_api"],
    "mega_triton_kernel.kernels": ["mega.kernels", "mega.task_api"],
    "mega_triton_kernel.kernels.task_context": ["mega.kernels", "mega.task_api"],
    "mega_triton_kernel.models.layers": ["mega.layers"],
    "mega_triton_kernel.models.layers.tp_attn": ["mega.layers"],
    "mega_triton_kernel.models.layers.tp_mlp": ["mega.layers"],
    "mega_triton_kernel.models.utils": ["mega.layers"],
    "mega_triton_kernel.models": ["mega.model_builder", "mega.qwen3"],
    "mega_triton_kernel.models.model_builder": ["mega.model_builder", "mega.task_api"],
    "mega_triton_kernel.models.qwen3": ["mega.qwen3"],
    "mega_triton_kernel.models.paged_kv_cache": ["models.kv_cache"],
    "mega_triton_kernel.tasks.utils": ["mega.tasks", "mega.task_api"],
    "mega_triton_kernel.kernels.utils": ["ops.common_ops"],
    "kernels.nvidia.gemm_rs_threadblock_swizzle": ["ops.threadblock_swizzle"],
    "kernels.nvidia.ag_gemm_threadblock_swizzle": ["ops.threadblock_swizzle"],
    "kernels.nvidia.threadblock_swizzle_ag_moe_triton": ["ops.threadblock_swizzle"],
}
_PACKAGES = {k for k in ALIASES if any(o.startswith(k + ".") for o in ALIASES)}


class _AliasLoader(importlib.abc.Loader):

    def __init__(self, key):
        self.key = key

    def create_module(self, spec):
        targets = ALIASES[self.key]
        if len(targets) == 1 and self.key not in _PACKAGES:
            return importlib.import_module(_P + targets[0])      # the very same module object
        mod = types.ModuleType(spec.name)
        mod.__doc__ = f"facade namespace over {', '.join(_P + t for t in targets) or 'sub-modules'}"
        for t in targets:
            src = importlib.import_module(_P + t)
            names = getattr(src, "__all__", None) or [n for n in vars(src) if not n.startswith("__")]
            for n in names:
                if not hasattr(mod, n) and hasattr(src, n):
                    setattr(mod, n, getattr(src, n))
        if self.key in _PACKAGES:
            mod.__path__ = []
        return mod

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("triton_dist."):
            return None
        key = fullname[len("triton_dist."):]
        if key not in ALIASES:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(key), is_package=key in _PACKAGES)


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
