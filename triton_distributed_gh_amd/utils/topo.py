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


# ---- PCIe / NUMA / xGMI link facts from sysfs (what the reference reads through NVML / nvidia-smi:
# python/triton_dist/utils.py:592-867).  Every probe degrades to None / a nominal value when the node of the
# device is not visible (containers without /sys/class/drm).
def _drm_device_dir(device_id: int = 0):
    """sysfs directory of GPU ``device_id`` (render-node order), or None."""
    import glob
    import os

    cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device")
                   if os.path.exists(os.path.join(d, "vendor")) and open(os.path.join(d, "vendor")).read().strip() == "0x1002")
    return cards[device_id] if 0 <= device_id < len(cards) else None


def _read(path: str):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def get_numa_node(device_id: int = 0):
    """NUMA node the GPU hangs off (reference name: ``get_numa_node_pynvml``), or None."""
    d = _drm_device_dir(device_id)
    v = _read(f"{d}/numa_node") if d else None
    try:
        node = int(v) if v not in (None, "") else -1
    except ValueError:
        node = -1
    return node if node >= 0 else None            # (the kernel reports -1 on single-node machines and in VMs)


def get_pcie_link_info(device_id: int = 0):
    """(GT/s per lane, lanes) of the GPU's PCIe link at its maximum, or None."""
    d = _drm_device_dir(device_id)
    if not d:
        return None
    speed, width = _read(f"{d}/max_link_speed"), _read(f"{d}/max_link_width")
    if not speed or not width:
        return None
    try:
        return float(speed.split()[0]), int(width)
    except ValueError:
        return None


def calculate_pcie_bandwidth(gt_per_lane: float, lanes: int) -> float:
    """GB/s per direction of a PCIe link: 8b/10b below 8 GT/s, 128b/130b from gen 3 on."""
    eff = 0.8 if gt_per_lane < 8.0 else 128.0 / 130.0
    return gt_per_lane * eff * lanes / 8.0


def get_pcie_link_max_speed(device_id: int = 0) -> float:
    """GB/s per direction of the GPU's PCIe link (nominal gen 5 x16 = 63 GB/s when sysfs is not visible)."""
    info = get_pcie_link_info(device_id)
    return calculate_pcie_bandwidth(*info) if info else calculate_pcie_bandwidth(32.0, 16)


def get_xgmi_link_max_speed(device_id: int = 0) -> float:
    """GB/s per direction of ONE xGMI link between two GPUs of an MI355X node (reference name:
    ``get_nvlink_max_speed``).  The platform figure is 153.6 GB/s per link, both directions together."""
    return 76.8


def get_active_xgmi_links(device_id: int = 0) -> int:
    """Peers this GPU reaches over xGMI (reference name: ``get_active_nvlinks_pynvml``): every other GPU of the
    node that it can access directly."""
    n = device_count()
    return sum(1 for p in range(n) if p != device_id and can_access_peer(device_id, p))


def device_count() -> int:
    import torch

    return torch.cuda.device_count() if torch.cuda.is_available() else 0


class RocmSmiUtil:
    """The node's GPU topology as ``rocm-smi`` reports it (reference: ``NvidiaSmiUtil``, python/triton_dist/
    utils.py:640-720, which parses ``nvidia-smi topo``).  Every method returns None when the tool is missing
    or prints something unexpected - callers fall back to the HIP peer-access probes above."""

    @staticmethod
    def _run(*args):
        import shutil
        import subprocess

        exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        try:
            res = subprocess.run([exe, *args, "--json"], capture_output=True, text=True, timeout=20)
        except (OSError, subprocess.TimeoutExpired):
            return None
        if res.returncode != 0:
            return None
        try:
            import json

            return json.loads(res.stdout)
        except ValueError:
            return None

    @staticmethod
    def get_link_types():
        """{(gpu i, gpu j): "XGMI" | "PCIE"} from ``rocm-smi --showtopotype``."""
        data = RocmSmiUtil._run("--showtopotype")
        if not isinstance(data, dict):
            return None
        out = {}
        for section in data.values():
            if not isinstance(section, dict):
                continue
            for key, val in section.items():
                # keys look like "(Topology) Link type between DRM devices 0 and 1"
                words = key.replace(",", " ").split()
                ids = [int(w) for w in words if w.isdigit()]
                if len(ids) >= 2 and isinstance(val, str):
                    out[(ids[-2], ids[-1])] = val.strip().upper()
        return out or None

    @staticmethod
    def get_gpu_numa_node(device_id: int = 0):
        return get_numa_node(device_id)

    @staticmethod
    def has_fullmesh_xgmi(num_gpus: int):
        links = RocmSmiUtil.get_link_types()
        if links is None:
            return None
        return all(links.get((i, j), links.get((j, i))) == "XGMI" for i in range(num_gpus) for j in range(i + 1, num_gpus))
