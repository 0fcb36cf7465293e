"""``triton_dist.utils`` under its reference names (reference: python/triton_dist/utils.py).

Everything here is a thin adapter over the native runtime: the symmetric heap replaces
NVSHMEM / rocSHMEM allocation (``nvshmem_create_tensor(s)``, utils.py:114-136), the IPC flag
barrier replaces ``nvshmem_barrier_all_on_stream`` (:162-164), stream memory operations replace
``nvshmem_signal_wait`` (:170-179), and the xGMI probes replace the NVLink / PCIe / NUMA probes
(:592-867)."""
from __future__ import annotations

import contextlib
import json
from typing import List, Optional

import torch

from ..ops.common_ops import BarrierAllContext, barrier_all_on_stream, set_signal, wait_eq  # noqa: F401
from . import distributed as _d
from .distributed import finalize_distributed, get_env, init_seed, is_initialized  # noqa: F401
from .perf import (_make_tensor, assert_allclose, bitwise_equal, dist_print, generate_data, perf_func,  # noqa: F401
                   sleep_async)
from .profile import get_torch_prof_ctx, group_profile, merge_traces  # noqa: F401
from .profile import ParallelJsonDumper, process_trace_json  # noqa: F401
from .topo import (HIP_CHECK, calculate_pcie_bandwidth, can_access_peer, device_props,  # noqa: F401
                   estimate_all_gather_time_ms, estimate_gemm_sol_time_ms, estimate_reduce_scatter_time_ms,
                   get_active_xgmi_links, get_intranode_max_speed_gbps, get_numa_node, get_numa_world_size,
                   get_pcie_link_info, get_pcie_link_max_speed, get_xgmi_link_max_speed, has_fullmesh_xgmi, has_tma,
                   RocmSmiUtil,
                   is_cuda, is_hip, is_nvshmem_multimem_supported, num_cus, num_xcds, requires,
                   supports_p2p_native_atomic)

NVSHMEM_SIGNAL_DTYPE = torch.int64      # (the AG-GEMM / GEMM-RS flag arrays are int32)
CUDA_CHECK = HIP_CHECK
has_fullmesh_nvlink = has_fullmesh_xgmi
# the reference's NVML / nvidia-smi probe names, served by the sysfs / HIP probes of utils/topo.py
has_fullmesh_nvlink_pynvml = has_fullmesh_xgmi
get_numa_node_pynvml = get_numa_node
get_nvlink_max_speed = get_nvlink_max_speed_pynvml = get_nvlink_max_speed_nvsmi = get_xgmi_link_max_speed
get_active_nvlinks_pynvml = get_active_xgmi_links
get_pcie_link_max_speed_pynvml = get_pcie_link_max_speed_nvsmi = get_pcie_link_max_speed
get_pcie_link_info_nvsmi = get_pcie_link_info
_merge_json = _merge_json_v1 = _merge_json_v2 = merge_traces
NvidiaSmiUtil = RocmSmiUtil


def parse_nvml_field_value(value):
    """Kept for source compatibility (the reference decodes NVML's tagged unions); sysfs values are plain text."""
    return value


def cuda_occupancy_max_activate_blocks_per_multiprocessor(kernel, num_warps: int, shared_memory: int = 0) -> int:
    """Resident workgroups per compute unit of a user kernel loaded with ``triton_dist.tools.load_module``
    (``num_warps`` counts 64-wide waves here)."""
    return kernel.max_active_blocks_per_cu(64 * int(num_warps), int(shared_memory))


def ensure_nvml_initialized() -> None:
    """Nothing to initialise: the probes read sysfs and the HIP runtime."""


def triton_packed_version() -> int:
    """The reference packs Triton's version into an int for AOT cache keys; there is no Triton here - the
    ROCm toolchain's HIP version plays that role."""
    v = getattr(torch.version, "hip", None) or "0.0"
    parts = [int("".join(c for c in x if c.isdigit()) or 0) for x in v.split(".")[:3]] + [0, 0]
    return (parts[0] << 16) | (parts[1] << 8) | (parts[2] & 0xFF)


def initialize_distributed(seed=None, **kwargs):
    """Bootstrap RCCL / gloo + the symmetric heap; returns the tensor-parallel process group like
    the reference (None in a single-process run).  ``get_env()`` gives the full environment."""
    env = _d.initialize_distributed(seed=seed, **kwargs)
    return env.tp_group


def init_nvshmem_by_torch_process_group(pg=None):
    """Kept for source compatibility: the heap is created by ``initialize_distributed``."""
    if not is_initialized():
        _d.initialize_distributed()


def nvshmem_create_tensor(shape, dtype) -> torch.Tensor:
    """Collective symmetric allocation: same offset in every rank's arena."""
    arena = "sig" if dtype in (torch.int32, torch.int64, torch.uint64) and _numel(shape) <= 4096 else "data"
    return get_env().heap.create_tensor(list(shape) if not isinstance(shape, int) else [shape], dtype, arena=arena)


def nvshmem_create_tensors(shape, dtype, rank=None, local_world_size=None) -> List[torch.Tensor]:
    """Per-peer views of one symmetric tensor (index ``rank`` is the local one)."""
    t = nvshmem_create_tensor(shape, dtype)
    return get_env().heap.peer_views(t)


def _numel(shape) -> int:
    if isinstance(shape, int):
        return shape
    n = 1
    for s in shape:
        n *= int(s)
    return n


def nvshmem_free_tensor_sync(tensor) -> None:
    env = get_env()
    if env.backend == "hip":
        torch.cuda.synchronize()
    env.heap.free_tensor(tensor)


# rocSHMEM-flavoured aliases (reference: shmem/rocshmem_bind/python/pyrocshmem/__init__.py:93-108)
rocshmem_create_tensor = nvshmem_create_tensor
rocshmem_create_tensor_list_intra_node = nvshmem_create_tensors


def nvshmem_barrier_all_on_stream(stream: Optional["torch.cuda.Stream"] = None) -> None:
    env = get_env()
    ctx = getattr(env, "_barrier_all_ctx", None)      # one per environment: virtual ranks each keep their own
    if ctx is None:
        ctx = env._barrier_all_ctx = BarrierAllContext(env=env)
    barrier_all_on_stream(ctx, stream)


def nvshmem_signal_wait(signal: torch.Tensor, pe: int, signal_val: int, signal_op: int = 0,
                        stream: Optional["torch.cuda.Stream"] = None) -> None:
    """Make ``stream`` wait for ``signal`` (this rank's copy) to reach ``signal_val``."""
    wait_eq(signal, signal_val, stream)


class TorchStreamWrapper:
    """Gives a torch stream the ``__cuda_stream__`` protocol some callers expect."""

    def __init__(self, pt_stream: "torch.cuda.Stream"):
        self.pt_stream = pt_stream
        self.handle = pt_stream.cuda_stream

    def __cuda_stream__(self):
        return (0, self.handle)


@contextlib.contextmanager
def with_torch_deterministic(mode: bool, warn_only: bool = True):
    old, old_warn = torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled()
    torch.use_deterministic_algorithms(mode, warn_only=warn_only)
    try:
        yield
    finally:
        torch.use_deterministic_algorithms(old, warn_only=old_warn)


def is_fp8_dtype(dtype: torch.dtype) -> bool:
    return dtype in (torch.float8_e4m3fn, torch.float8_e5m2, torch.float8_e4m3fnuz, torch.float8_e5m2fnuz)


def load_json(json_file):
    with open(json_file, "r", encoding="utf-8", errors="replace") as f:
        return json.load(f)


def requires_p2p_native_atomic(fn):
    def wrapper(*args, **kwargs):
        if not supports_p2p_native_atomic():
            import warnings

            warnings.warn(f"{fn.__name__} uses peer atomics that this system does not support natively")
        return fn(*args, **kwargs)

    return wrapper


def get_device_max_shared_memory_size(device=0) -> int:
    """LDS bytes a workgroup can use (160 KiB on gfx950)."""
    try:
        return int(device_props(device if isinstance(device, int) else 0).get("shared_mem_per_block", 163840))
    except Exception:
        return 163840


def get_device_property(device_id=0):
    return torch.cuda.get_device_properties(device_id)


def get_intranode_max_speed(gpu_index=0, with_scale: bool = False):
    return get_intranode_max_speed_gbps(get_env().world_size if is_initialized() else 8)


def get_numa_node(gpu_index=0) -> int:
    try:
        with open(f"/sys/class/drm/card{gpu_index}/device/numa_node") as f:
            return max(int(f.read().strip()), 0)
    except Exception:
        return 0


def support_launch_cooperative_grid() -> bool:
    return True


def launch_cooperative_grid_options():
    """Persistent kernels here size their grid to the CU count and bound every spin; no special
    launch attribute is needed."""
    return {}


def get_nvshmem_home():
    return None


def has_nvshmemi_bc_built() -> bool:
    return False
