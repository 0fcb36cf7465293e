"""Speed-of-light estimates for GEMMs and collectives on an MI355X node.

Role of the reference's kernels/nvidia/gemm_perf_model.py:61-236 (tensor-core TFLOPS / DRAM tables,
``estimate_gemm_sol_time_ms``) and comm_perf_model.py:35-116 (NIC discovery,
``estimate_all_gather_time_ms`` / ``estimate_reduce_scatter_time_ms``), with the tables replaced by
the MI355X numbers: dense MFMA peaks (no 2:1 sparsity), HBM3E bandwidth and the xGMI mesh - seven
links per GPU, one per peer, so a full-mesh collective moves (w-1) shards over (w-1) links at once
and is bounded by ONE link's bandwidth per shard, not by a ring of hops."""
from __future__ import annotations

import torch

# dense peaks, TFLOP/s
MI355X_TFLOPS = {
    torch.bfloat16: 2500.0, torch.float16: 2500.0, torch.float8_e4m3fn: 5000.0, torch.float8_e5m2: 5000.0,
    torch.int8: 5000.0, torch.float32: 157.0,
}
MI355X_HBM_GBPS = 8000.0
XGMI_LINK_GBPS = 153.0          # per direction, per peer link
XGMI_LINKS_PER_GPU = 7


def get_max_tensorcore_tflops(dtype: torch.dtype = torch.bfloat16, device_id: int = 0) -> float:
    return MI355X_TFLOPS.get(dtype, 2500.0)


get_max_mfma_tflops = get_max_tensorcore_tflops


def get_dram_gbps(device_id: int = 0) -> float:
    return MI355X_HBM_GBPS


def get_tensorcore_tflops_by_device_name(name: str = "MI355X", dtype: torch.dtype = torch.bfloat16) -> float:
    return get_max_tensorcore_tflops(dtype)


def estimate_gemm_sol_time_ms(M: int, N: int, K: int, dtype: torch.dtype = torch.bfloat16) -> float:
    """max(compute bound, memory bound) for C[M,N] = A[M,K] B[N,K]^T."""
    esize = torch.empty((), dtype=dtype).element_size()
    t_flop = 2.0 * M * N * K / (get_max_tensorcore_tflops(dtype) * 1e12)
    t_mem = (M * K + N * K + M * N) * esize / (MI355X_HBM_GBPS * 1e9)
    return max(t_flop, t_mem) * 1e3


def get_nic_gbps_per_gpu() -> float:
    """No NIC is involved inside one node; reported for API compatibility (0 = not discovered)."""
    return 0.0


def intranode_egress_gbps(world_size: int) -> float:
    """Aggregate egress of one GPU in a full-mesh collective over ``world_size`` ranks."""
    return XGMI_LINK_GBPS * max(min(world_size - 1, XGMI_LINKS_PER_GPU), 1)


def estimate_all_gather_time_ms(nbytes_per_rank: int, world_size: int, local_world_size: int = 8,
                                efficiency: float = 0.85) -> float:
    """Full-mesh push: every rank sends its shard to w-1 peers on w-1 links in parallel."""
    if world_size <= 1:
        return 0.0
    return nbytes_per_rank / (XGMI_LINK_GBPS * 1e9 * efficiency) * 1e3


def estimate_reduce_scatter_time_ms(nbytes_total: int, world_size: int, local_world_size: int = 8,
                                    efficiency: float = 0.85) -> float:
    """One-hop pull-reduce: every rank reads its 1/w segment from w-1 peers in parallel."""
    if world_size <= 1:
        return 0.0
    return (nbytes_total / world_size) / (XGMI_LINK_GBPS * 1e9 * efficiency) * 1e3


def estimate_all_reduce_time_ms(nbytes: int, world_size: int, efficiency: float = 0.85) -> float:
    """Two-shot (reduce-scatter + all-gather) bandwidth bound."""
    return 2.0 * estimate_reduce_scatter_time_ms(nbytes, world_size, efficiency=efficiency)
