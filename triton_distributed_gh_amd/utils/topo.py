"""Hardware probes for an MI355X node: xGMI mesh, P2P atomics, occupancy, perf models.

Replaces the reference's NVLink / PCIe / NUMA / NIC probes and SOL models
(reference: python/triton_dist/utils.py:592-1048 - ``has_fullmesh_nvlink``,
``get_intranode_max_speed``, ``supports_p2p_native_atomic``, ``is_nvshmem_multimem_supported``,
``has_tma``; kernels/nvidia/comm_perf_model.py:94-116; kernels/nvidia/gemm_perf_model.py:61-236).
There is no NVLS / TMA on this hardware, so those probes answer False and the collectives that the
reference builds on them are mapped onto full-mesh peer loads/stores.
"""
from __future__ import annotations

import ctypes
import functools
import os
from typing import Optional

import torch

from .. import _lib

# Measured / datasheet constants for MI355X (gfx950).
MI355X_CUS = 256
MI355X_XCDS = 8
MI355X_HBM_TBPS = 6.3            # achievable streaming bandwidth, TB/s
MI355X_DENSE_TFLOPS = {torch.bfloat16: 2500.0, torch.float16: 2500.0, torch.float8_e4m3fn: 5000.0,
                       torch.float32: 157.3, torch.int8: 5000.0}
XGMI_LINK_GBPS_PER_DIR = 76.8    # 153.6 GB/s bidirectional per link
XGMI_LINKS_PER_GPU = 7


def is_hip() -> bool:
    return torch.cuda.is_available() and getattr(torch.version, "hip", None) is not None


def is_cuda() -> bool:
    return False


@functools.lru_cache(maxsize=None)
def device_props(dev: int = 0) -> dict:
    if not torch.cuda.is_available():
        return {"cus": 0, "lds_per_cu": 0, "clock_khz": 0, "mem_mib": 0, "wave": 64, "l2": 0, "arch": "cpu"}
    out = (ctypes.c_longlong * 6)()
    name = ctypes.create_string_buffer(64)
    _lib.check(_lib.hip().tdist_device_props(dev, out, name, 64), "tdist_device_props")
    return {"cus": int(out[0]), "lds_per_cu": int(out[1]), "clock_khz": int(out[2]), "mem_mib": int(out[3]),
            "wave": int(out[4]), "l2": int(out[5]), "arch": name.value.decode()}


def num_cus(dev: int = 0) -> int:
    return device_props(dev)["cus"] or MI355X_CUS


def num_xcds() -> int:
    return MI355X_XCDS


def p2p_attribute(attr: int, src: int, dst: int) -> int:
    out = ctypes.c_int(0)
    _lib.check(_lib.hip().tdist_p2p_attribute(attr, src, dst, ctypes.byref(out)), "hipDeviceGetP2PAttribute")
    return out.value


def can_access_peer(src: int, dst: int) -> bool:
    out = ctypes.c_int(0)
    _lib.check(_lib.hip().tdist_can_access_peer(src, dst, ctypes.byref(out)), "hipDeviceCanAccessPeer")
    return bool(out.value)


def has_fullmesh_xgmi(world_size: Optional[int] = None) -> bool:
    """True when every pair of visible GPUs has direct peer access (the xGMI mesh of one node)."""
    if not torch.cuda.is_available():
        return False
    n = world_size or torch.cuda.device_count()
    n = min(n, torch.cuda.device_count())
    return all(can_access_peer(i, j) for i in range(n) for j in range(n) if i != j)


# name kept for scripts written against the reference
has_fullmesh_nvlink = has_fullmesh_xgmi


def supports_p2p_native_atomic() -> bool:
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        return torch.cuda.is_available()
    return bool(p2p_attribute(2, 0, 1))


def is_nvshmem_multimem_supported() -> bool:
    return False   # no multicast / in-switch reduction on xGMI


def has_tma() -> bool:
    return False


def get_numa_world_size() -> int:
    return torch.cuda.device_count() if torch.cuda.is_available() else 1


def get_intranode_max_speed_gbps(world_size: int) -> float:
    """Aggregate per-GPU egress over the links a world of `world_size` GPUs can use, GB/s per direction."""
    links = max(1, min(world_size - 1, XGMI_LINKS_PER_GPU))
    return links * XGMI_LINK_GBPS_PER_DIR


# ---- speed-of-light models ---------------------------------------------------------------------
def estimate_gemm_sol_time_ms(M: int, N: int, K: int, dtype=torch.bfloat16) -> float:
    flops = 2.0 * M * N * K
    esize = torch.empty((), dtype=dtype).element_size()
    compute = flops / (MI355X_DENSE_TFLOPS.get(dtype, 2500.0) * 1e12)
    memory = (M * K + N * K + M * N) * esize / (MI355X_HBM_TBPS * 1e12)
    return max(compute, memory) * 1e3


def estimate_all_gather_time_ms(nbytes_per_rank: int, world_size: int, efficiency: float = 0.8) -> float:
    """Full-mesh push: every rank sends its shard once over each of its w-1 links concurrently."""
    if world_size <= 1:
        return 0.0
    return nbytes_per_rank / (XGMI_LINK_GBPS_PER_DIR * 1e9 * efficiency) * 1e3


def estimate_reduce_scatter_time_ms(nbytes_total: int, world_size: int, efficiency: float = 0.8) -> float:
    """Full-mesh scatter: each rank sends nbytes_total / w to each peer, one peer per link."""
    if world_size <= 1:
        return 0.0
    return (nbytes_total / world_size) / (XGMI_LINK_GBPS_PER_DIR * 1e9 * efficiency) * 1e3


def requires(cond_fn):
    """Decorator: skip the wrapped function with a message when the hardware lacks a feature."""

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            if not cond_fn():
                raise RuntimeError(f"{fn.__name__} requires {cond_fn.__name__}() on this machine")
            return fn(*a, **k)

        return wrapped

    return deco


def HIP_CHECK(err) -> None:
    """Accepts an int error code or a tuple whose first item is the code (hip-python convention)."""
    code = err[0] if isinstance(err, tuple) else err
    code = int(getattr(code, "value", code))
    if code != 0:
        raise RuntimeError(f"HIP error {code}")


CUDA_CHECK = HIP_CHECK
