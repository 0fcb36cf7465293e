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


# ---- what this repo's kernels reach on one MI355X (profiles/: GEMM 4096 x 8192 x 8192 at 1.55-1.62 PFLOP/s,
# flash decode and the decode GEMM at 5.3-5.6 TB/s, one pushing workgroup at 12-16 GB/s) -------------------
ACHIEVED = {
    "gemm_tflops_bf16": 1580.0,        # large-tile bf16 GEMM, MFMA-bound shapes
    "hbm_stream_gbps": 5500.0,         # streaming kernels (weights / KV read once)
    "push_wg_gbps": 14.0,              # one 512-thread workgroup storing write-through rows to a peer
    "launch_us": 4.5,                  # dependent kernel boundary inside a HIP graph
    "flag_hop_us": 3.6,                # producer flag -> consumer wake-up across CUs
}


def estimate_gemm_time_ms(M: int, N: int, K: int, dtype: torch.dtype = torch.bfloat16, num_cus: int = 256) -> float:
    """What the tile engine is expected to take (not the speed of light): achieved MFMA rate scaled by
    the wave quantisation of 256 x 256 tiles over ``num_cus`` workgroups, or the streaming bound when the
    operands dominate (decode shapes)."""
    esize = torch.empty((), dtype=dtype).element_size()
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    # fewer tiles than CUs: K is cut across workgroups until the chip is full (at least 512 deep per split)
    splits = max(1, min(num_cus // tiles, K // 512)) if tiles < num_cus else 1
    rounds = (tiles * splits + num_cus - 1) // num_cus
    per_cu = ACHIEVED["gemm_tflops_bf16"] * (get_max_tensorcore_tflops(dtype) / 2500.0) * 1e12 / 256.0
    t_flop = rounds * 2.0 * 256 * 256 * (K / splits) / per_cu
    t_mem = (M * K + N * K + M * N) * esize / (ACHIEVED["hbm_stream_gbps"] * 1e9)
    return max(t_flop, t_mem) * 1e3


def estimate_ag_gemm_time_ms(M: int, N_per_rank: int, K: int, world_size: int, dtype: torch.dtype = torch.bfloat16,
                             push_wgs: int = 0, num_cus: int = 256) -> dict:
    """AG + GEMM with the all-gather overlapped tile by tile: the slower of the two pipelines plus the part
    that cannot overlap (the first chunk has to land before any foreign tile starts).  ``push_wgs`` in-kernel
    producer workgroups are taken from the GEMM.  Returns the parts in ms."""
    esize = torch.empty((), dtype=dtype).element_size()
    shard = (M // max(world_size, 1)) * K * esize
    t_ag = estimate_all_gather_time_ms(shard, world_size)
    t_gemm = estimate_gemm_time_ms(M, N_per_rank, K, dtype, max(num_cus - push_wgs, 1))
    t_push = 0.0
    if push_wgs > 0 and world_size > 1:
        t_push = shard * (world_size - 1) / (push_wgs * ACHIEVED["push_wg_gbps"] * 1e9) * 1e3
    head = min(t_ag, max(t_ag, t_push) / max(M // max(world_size, 1) // 256, 1))      # one 256-row chunk
    total = max(t_gemm, t_ag, t_push) + (head if world_size > 1 else 0.0)
    return {"gemm_ms": t_gemm, "all_gather_ms": t_ag, "push_ms": t_push, "exposed_ms": head, "total_ms": total}


def estimate_gemm_rs_time_ms(M: int, N: int, K_per_rank: int, world_size: int, dtype: torch.dtype = torch.bfloat16,
                             num_cus: int = 256) -> dict:
    """GEMM + reduce-scatter, scatter of row chunk c behind the GEMM of chunk c + 1: the slower pipeline plus
    the last chunk's scatter and reduce, which nothing hides."""
    esize = torch.empty((), dtype=dtype).element_size()
    t_gemm = estimate_gemm_time_ms(M, N, K_per_rank, dtype, num_cus)
    t_rs = estimate_reduce_scatter_time_ms(M * N * esize, world_size)
    chunks = max(M // max(world_size, 1) // 256, 1) * max(world_size, 1)
    tail = t_rs / chunks + (M // max(world_size, 1)) * N * esize * (world_size + 1) / (ACHIEVED["hbm_stream_gbps"] * 1e9) * 1e3 / max(chunks // max(world_size, 1), 1)
    total = max(t_gemm, t_rs) + (tail if world_size > 1 else 0.0)
    return {"gemm_ms": t_gemm, "reduce_scatter_ms": t_rs, "exposed_ms": tail if world_size > 1 else 0.0, "total_ms": total}


def suggest_push_workgroups(M: int, N_per_rank: int, K: int, world_size: int, dtype: torch.dtype = torch.bfloat16,
                            num_cus: int = 256, max_fraction: float = 0.25) -> int:
    """Producer workgroups that keep the all-gather off the critical path: enough to move the shard to every
    peer within the GEMM's time (or to saturate the links, whichever is less), a multiple of the number of
    peers, never more than ``max_fraction`` of the CUs - past that the copy engines are the better producer."""
    if world_size <= 1:
        return 0
    esize = torch.empty((), dtype=dtype).element_size()
    peers = world_size - 1
    out_bytes = (M // world_size) * K * esize * peers
    t = max(estimate_gemm_time_ms(M, N_per_rank, K, dtype, num_cus), estimate_all_gather_time_ms(out_bytes // peers, world_size))
    need = out_bytes / (t * 1e-3) / (ACHIEVED["push_wg_gbps"] * 1e9)
    n = int(-(-need // peers)) * peers
    return max(peers, min(n, int(num_cus * max_fraction) // peers * peers))


def prefer_copy_engine_producer(M: int, N_per_rank: int, K: int, world_size: int,
                                dtype: torch.dtype = torch.bfloat16, num_cus: int = 256) -> bool:
    """True where the in-kernel producer would need more than a quarter of the CUs to keep up: the all-gather
    is then cheaper on the SDMA engines, which take nothing from the GEMM."""
    if world_size <= 1:
        return False
    esize = torch.empty((), dtype=dtype).element_size()
    peers = world_size - 1
    out_bytes = (M // world_size) * K * esize * peers
    t = estimate_gemm_time_ms(M, N_per_rank, K, dtype, num_cus)
    need = out_bytes / (t * 1e-3) / (ACHIEVED["push_wg_gbps"] * 1e9)
    return need > num_cus * 0.25


def estimate_decode_step_ms(weight_bytes: int, kv_bytes: int, launches: int) -> float:
    """A decode step streams the weights and the attended KV once and pays a kernel boundary per launch."""
    return ((weight_bytes + kv_bytes) / (ACHIEVED["hbm_stream_gbps"] * 1e9) + launches * ACHIEVED["launch_us"] * 1e-6) * 1e3
