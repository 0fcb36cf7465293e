"""Entry points the reference exposes per ALGORITHM VARIANT, served by this package's one implementation each.

The reference keeps one host function per variant of a collective - ring / full-mesh / NUMA-2D / multimem /
TMA forms of all-gather, all-reduce and reduce-scatter (python/triton_dist/kernels/nvidia/allgather.py:81-560,
low_latency_allgather.py:819-960, allreduce.py:333-820, reduce_scatter.py:150-870), Triton / torch forms of the
MoE index helpers (moe_utils.py:40-330) and device-table look-ups of the performance models
(gemm_perf_model.py:30-230, comm_perf_model.py:30-120).  On an MI355X node every pair of GPUs has its own xGMI
link, so the full-mesh algorithm is the right one at every size and the variants collapse onto it; the
functions below keep the reference's names and argument lists and say which implementation runs.
Nothing here adds an algorithm: it is the compatibility surface of ``triton_dist.kernels.nvidia.*``.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from ..utils.distributed import get_env
from . import allgather as _ag
from . import allreduce as _ar
from . import moe as _moe
from . import reduce_scatter as _rs
from .compat_nvidia import cp_engine_producer_all_gather_intra_node


# ---- copy-engine all-gather producers (allgather.py:81-560) -------------------------------------------------
def cp_engine_producer_all_gather_full_mesh_push(rank, num_ranks, local_tensor: torch.Tensor,
                                                 remote_tensor_buffers: List[torch.Tensor],
                                                 barrier_buffers: List[torch.Tensor], stream=None, **_):
    """Every rank copies its shard into slot ``rank`` of every peer's buffer on the copy engines, then raises
    that peer's flag ``rank`` in stream order."""
    return cp_engine_producer_all_gather_intra_node(rank, num_ranks, local_tensor, remote_tensor_buffers, barrier_buffers,
                                                    stream)


def cp_engine_producer_all_gather_full_mesh_pull(rank, num_ranks, local_tensor: torch.Tensor,
                                                 remote_tensor_buffers: List[torch.Tensor],
                                                 barrier_buffers: List[torch.Tensor], stream=None, **_):
    """Pull form: this rank copies every PEER's slot out of that peer's buffer into its own and raises its own
    flags.  The peers must have published their slot (slot ``p`` of ``remote_tensor_buffers[p]``) beforehand."""
    from .common_ops import memcpy_async, set_signal

    env = get_env()
    m = local_tensor.shape[0]
    nbytes = local_tensor.numel() * local_tensor.element_size()
    mine = remote_tensor_buffers[rank]
    mine[rank * m:(rank + 1) * m].copy_(local_tensor)
    for i in range(1, num_ranks):
        peer = (rank + i) % num_ranks
        src = remote_tensor_buffers[peer][peer * m:(peer + 1) * m]
        dst = mine[peer * m:(peer + 1) * m]
        if env.backend == "host":
            dst.copy_(src)
        else:
            memcpy_async(dst, src, nbytes, stream)
        if barrier_buffers is not None:
            set_signal(barrier_buffers[rank][peer:peer + 1], 1, stream)


# ring and NUMA-aware orders exist for NVLink-switch-less or multi-socket topologies; one hop over a dedicated
# xGMI link per peer beats any of them here
cp_engine_producer_all_gather_ring_push_1d = cp_engine_producer_all_gather_full_mesh_push
cp_engine_producer_all_gather_ring_push_numa_2d = cp_engine_producer_all_gather_full_mesh_push


def cp_engine_producer_all_gather_ring_push_2d_inter_node(rank, num_local_ranks, num_ranks, local_tensor, remote_tensor_buffers,
                                                          barrier_buffers, stream=None, **_):
    assert num_ranks == num_local_ranks, "one node: the inter-node form runs the intra-node full mesh"
    return cp_engine_producer_all_gather_full_mesh_push(rank, num_ranks, local_tensor, remote_tensor_buffers, barrier_buffers,
                                                        stream)


def cp_engine_producer_all_gather_full_mesh_pull_inter_node(rank, local_world_size, world_size, local_tensor, ag_buffer,
                                                            signal_buffer, stream=None, **_):
    assert world_size == local_world_size, "one node: the inter-node form runs the intra-node full mesh"
    return cp_engine_producer_all_gather_full_mesh_pull(rank, world_size, local_tensor, ag_buffer, signal_buffer, stream)


# ---- low-latency all-gather, one function per mode (low_latency_allgather.py:819-960) ----------------------
def _fast(mode):
    def run(ctx: "_ag.FastAllGatherContext", symm_buffer: torch.Tensor):
        return _ag.fast_allgather(symm_buffer, ctx, mode=mode)

    run.__name__ = f"fast_allgather_{mode}"
    run.__doc__ = f"``fast_allgather(symm_buffer, ctx, mode={mode!r})`` (ops/allgather.py)."
    return run


fast_allgather_pull = _fast("pull")
fast_allgather_push_2d = _fast("push_2d")
fast_allgather_push_3d = _fast("push_3d")
fast_allgather_push_2d_ll = _fast("push_2d_ll")
fast_allgather_push_2d_ll_multimem = _fast("push_2d_ll_multimem")
fast_allgather_push_numa_2d = _fast("push_numa_2d")
fast_allgather_push_numa_2d_ll = _fast("push_numa_2d_ll")
fast_allgather_push_numa_2d_ll_multinode = _fast("push_2d_ll_multinode")


# ---- all-reduce, one function per method (allreduce.py:333-820) ----------------------------------------------
def _allreduce(method: "_ar.AllReduceMethod"):
    def run(ctx: "_ar.AllReduceContext", x: torch.Tensor, output: Optional[torch.Tensor] = None, max_sm: int = -1, **_):
        return _ar.all_reduce(x, method, ctx, output, max_sm)

    run.__doc__ = f"``all_reduce(x, AllReduceMethod.{method.name}, ctx)`` (ops/allreduce.py)."
    return run


allreduce_one_shot_push_intra_node = _allreduce(_ar.AllReduceMethod.OneShot)
allreduce_one_shot_tma_push_intra_node = _allreduce(_ar.AllReduceMethod.OneShot_TMA)
allreduce_one_shot_multimem_intra_node = _allreduce(_ar.AllReduceMethod.OneShot_Multimem)
allreduce_two_shot_push_intra_node = _allreduce(_ar.AllReduceMethod.TwoShot)
allreduce_two_shot_multimem_intra_node = _allreduce(_ar.AllReduceMethod.TwoShot_Multimem)
allreduce_two_shot_multimem_st_intra_node = _allreduce(_ar.AllReduceMethod.TwoShot_Multimem_ST)
allreduce_double_tree_intra_node = _allreduce(_ar.AllReduceMethod.DoubleTree)


def get_tree_parent_and_children(N: int, rank: int):
    """Parent and children of ``rank`` in the two complementary binary trees of a double-tree all-reduce over
    ``N`` ranks (N a power of two): ((parent0, child0a, child0b), (parent1, child1a, child1b)), -1 = none.
    Tree 0 is the in-order binary tree over ranks 1 .. N-1 rooted at N/2 with rank 0 hung below rank 1;
    tree 1 is the same tree with every rank shifted by one (mod N), so a rank that is a leaf in one tree
    is an inner node in the other.  (The all-reduce kernels here are one-hop full-mesh; this helper serves code
    that plans tree schedules.)"""
    if N <= 0 or N & (N - 1):
        raise ValueError("N must be a power of two")

    def tree(r):
        if N == 1:
            return (-1, -1, -1)
        if r == 0:
            return (1 if N > 1 else -1, -1, -1)
        low = r & -r                              # lowest set bit = height of r in the in-order tree
        parent = -1 if r == N // 2 else ((r - low) | (low << 1))
        if low == 1:
            kids = [0] if r == 1 else []
        else:
            kids = [r - (low >> 1), r + (low >> 1)]
        kids += [-1] * (2 - len(kids))
        return (parent, kids[0], kids[1])

    t0 = tree(rank)
    s = tree((rank - 1) % N)
    t1 = tuple(-1 if v < 0 else (v + 1) % N for v in s)
    return t0, t1


# ---- reduce-scatter variants (reduce_scatter.py:150-870) ----------------------------------------------------
def _rs_variant(doc):
    def run(input: torch.Tensor, ctx: "_rs.ReduceScatter2DContext", output: Optional[torch.Tensor] = None, **_):
        return _rs.reduce_scatter_2d_op(input, ctx, output)

    run.__doc__ = doc + "  Runs ``reduce_scatter_2d_op`` (ops/reduce_scatter.py): every rank pushes the row block of " \
                        "each peer into that peer's scatter buffer over its own link and reduces what it received."
    return run


reduce_scatter_ring_push_1d_intra_node_ce = _rs_variant("Ring push on the copy engines in the reference.")
reduce_scatter_ring_push_1d_intra_node_sm = _rs_variant("Ring push from compute units in the reference.")
reduce_scatter_ring_push_1d_intra_node_sm_rma = _rs_variant("Ring push through SHMEM puts in the reference.")
reduce_scatter_for_each_node_ring = _rs_variant("Per-node ring in the reference.")
intra_node_scatter = _rs_variant("Scatter phase of the two-dimensional algorithm in the reference.")
ring_reduce_non_tma = ring_reduce_tma = _rs.ring_reduce


def add_continuous(lhs: torch.Tensor, rhs: torch.Tensor, output: Optional[torch.Tensor] = None, **_) -> torch.Tensor:
    """out = lhs + rhs over flat contiguous buffers, accumulated in fp32."""
    out = torch.empty_like(lhs) if output is None else output
    out.copy_((lhs.float() + rhs.float()).to(out.dtype))
    return out


def padded_to_BLOCK_M(input: torch.Tensor, world_size: int, BLOCK_SIZE_M: int) -> torch.Tensor:
    """Pad every rank's row block of ``input`` [M, N] to a multiple of ``BLOCK_SIZE_M`` rows (zeros)."""
    M, N = input.shape
    per = M // world_size
    padded = (per + BLOCK_SIZE_M - 1) // BLOCK_SIZE_M * BLOCK_SIZE_M
    if padded == per:
        return input
    out = torch.zeros(world_size, padded, N, dtype=input.dtype, device=input.device)
    out[:, :per].copy_(input.view(world_size, per, N))
    return out.view(world_size * padded, N)


# ---- MoE index helpers (moe_utils.py:40-330) ------------------------------------------------------------------
def histogram_by_expert_torch(chosen_experts: torch.Tensor, num_experts: int) -> torch.Tensor:
    return _moe.histogram(chosen_experts, num_experts)


histogram_by_expert_triton = histogram_by_expert_torch


def calc_scatter_index_torch(chosen_experts: torch.Tensor, num_experts: int = 0) -> torch.Tensor:
    """Position of every (token, k) pair in the expert-sorted order (stable)."""
    return _moe.calc_gather_scatter_index(chosen_experts, num_experts or int(chosen_experts.max()) + 1)[1]


def calc_gather_index_from_scatter_index(scatter_index: torch.Tensor, row_start: int = 0, row_end: Optional[int] = None):
    """Inverse permutation of ``scatter_index``, restricted to sorted rows [row_start, row_end)."""
    flat = scatter_index.reshape(-1).long()
    gather = torch.empty_like(flat)
    gather[flat] = torch.arange(flat.numel(), device=flat.device)
    return gather[row_start:row_end].int()


def calc_gather_index_torch(chosen_experts: torch.Tensor, num_experts: int = 0, row_start: int = 0,
                            row_end: Optional[int] = None) -> torch.Tensor:
    g = _moe.calc_gather_scatter_index(chosen_experts, num_experts or int(chosen_experts.max()) + 1)[0]
    return g[row_start:row_end]


def calc_gather_scatter_index_triton(chosen_experts: torch.Tensor, num_experts: int, **_):
    return _moe.calc_gather_scatter_index(chosen_experts, num_experts)


def calc_sorted_gather_index(chosen_experts: torch.Tensor, num_experts: int, world_size: int = 1, rank: int = 0, **_):
    """The expert-sorted, 256-row padded gather index of AG + grouped GEMM (``moe_align_block_size``)."""
    return _moe.moe_align_block_size(chosen_experts, num_experts, world_size, rank).sorted_ids


def reduce_topk_non_tma(x: torch.Tensor, topk: int, weights: Optional[torch.Tensor] = None, **_) -> torch.Tensor:
    return _moe.reduce_topk(x, topk, weights)


reduce_topk_tma = reduce_topk_non_tma


def moe_gather_rs_grouped_gemm(x, weights, chosen_experts, expert_weight, ctx, **kw):
    """Grouped GEMM + top-k reduce + reduce-scatter in one call (``run_moe_reduce_rs``)."""
    return _moe.run_moe_reduce_rs(x, weights, chosen_experts, expert_weight, ctx, n_chunks=kw.get("n_chunks"))


reduce_topk_reduce_scatter_intra_node = reduce_topk_reduce_scatter_a2a_intra_node = moe_gather_rs_grouped_gemm
reduce_topk_reduce_scatter_ring_intra_node = moe_gather_rs_grouped_gemm


def dtype_size_in_bytes(dtype: torch.dtype) -> int:
    return torch.empty((), dtype=dtype).element_size()


# ---- performance-model look-ups (gemm_perf_model.py:30-230, comm_perf_model.py:30-120) ----------------------
def is_fp8_dtype(dtype: torch.dtype) -> bool:
    return dtype in tuple(getattr(torch, n) for n in ("float8_e4m3fn", "float8_e5m2", "float8_e4m3fnuz", "float8_e5m2fnuz")
                          if hasattr(torch, n))


def get_device_property(device_id: int = 0):
    return torch.cuda.get_device_properties(device_id)


def get_device_multi_processor_count(device_id: int = 0) -> int:
    from ..utils.topo import num_cus

    return num_cus(device_id)


def get_clock_rate_in_khz(device_id: int = 0) -> int:
    from ..utils.topo import device_props

    return int(device_props(device_id).get("clock_khz") or 2_400_000)


def get_tensorcore_dtype_support(device_id: int = 0):
    """Input types of the MFMA units of gfx950."""
    return [torch.float16, torch.bfloat16, torch.float32, torch.int8] + [getattr(torch, n) for n in
                                                                          ("float8_e4m3fn", "float8_e5m2") if hasattr(torch, n)]


def get_tensorcore_tflops_by_calc(dtype: torch.dtype = torch.bfloat16, device_id: int = 0) -> float:
    """Dense MFMA peak from the device's own numbers: FLOP per clock per compute unit x compute units x clock
    (bf16 / fp16: 4 SIMDs x 1024 FLOP per clock; fp8 twice, fp32 one sixteenth of that)."""
    per_cu = {torch.bfloat16: 4096, torch.float16: 4096, torch.float32: 256, torch.int8: 8192}.get(dtype, 8192 if is_fp8_dtype(dtype) else 4096)
    return per_cu * get_device_multi_processor_count(device_id) * get_clock_rate_in_khz(device_id) * 1e3 / 1e12


def get_tensorcore_tflops(dtype: torch.dtype = torch.bfloat16, device_id: int = 0) -> float:
    from ..utils.perf_model import get_max_tensorcore_tflops

    return get_max_tensorcore_tflops(dtype, device_id)


def get_simd_tflops(dtype: torch.dtype = torch.float32, device_id: int = 0) -> float:
    """Vector-ALU peak: 4 SIMDs x 64 lanes x 2 FLOP per clock per compute unit (packed fp16 / bf16: twice)."""
    per_cu = 512 * (2 if dtype in (torch.float16, torch.bfloat16) else 1)
    return per_cu * get_device_multi_processor_count(device_id) * get_clock_rate_in_khz(device_id) * 1e3 / 1e12


get_tflops_approx = get_full_tflops_approx = get_tensorcore_tflops


def get_dram_gbps_by_device_name(name: str = "MI355X") -> float:
    from ..utils.perf_model import get_dram_gbps

    return get_dram_gbps()


def get_network_interfaces():
    """Names of the host's network interfaces that are up (the reference sizes inter-node transfers by them)."""
    import os

    out = []
    for name in sorted(os.listdir("/sys/class/net")) if os.path.isdir("/sys/class/net") else []:
        try:
            if name != "lo" and open(f"/sys/class/net/{name}/operstate").read().strip() == "up":
                out.append(name)
        except OSError:
            continue
    return out


def get_max_nic_bandwidth_gpbs() -> float:
    """Fastest NIC of the host in Gbit/s (0 when none reports a speed)."""
    best = 0.0
    for name in get_network_interfaces():
        try:
            best = max(best, float(open(f"/sys/class/net/{name}/speed").read().strip()) / 1000.0)
        except (OSError, ValueError):
            continue
    return best


# ---- AG + GEMM / GEMM + RS: the stages the reference exposes separately --------------------------------------
def matmul_get_configs():
    """The tile shapes the native GEMM engine is built with (the reference returns its Triton autotune space,
    kernels/nvidia/allgather_gemm.py:396 / kernels/amd/all_gather_gemm.py:102)."""
    return [{"BLOCK_SIZE_M": 256, "BLOCK_SIZE_N": 256, "BLOCK_SIZE_K": 64, "waves": 8, "stages": 2},
            {"BLOCK_SIZE_M": 128, "BLOCK_SIZE_N": 256, "BLOCK_SIZE_K": 64, "waves": 8, "stages": 2},
            {"BLOCK_SIZE_M": 16, "BLOCK_SIZE_N": 64, "BLOCK_SIZE_K": 32, "waves": 4, "stages": 2}]


get_hip_autotune_config = matmul_get_configs


def update_triton_config(M, N, K, dtype, world_size, local_world_size, config=None):
    """Workgroups to give the communication side of GEMM + RS for this shape (the reference rewrites a Triton
    config from the same two estimates, kernels/nvidia/gemm_reduce_scatter.py:482-506)."""
    from ..utils.perf_model import estimate_gemm_time_ms, estimate_reduce_scatter_time_ms

    gemm_ms = estimate_gemm_time_ms(M, N, K // max(1, world_size), dtype)
    rs_ms = estimate_reduce_scatter_time_ms(M * N * dtype_size_in_bytes(dtype), world_size, local_world_size)
    cfg = dict(config or matmul_get_configs()[0])
    cfg["comm_bound"] = rs_ms > gemm_ms
    cfg["gemm_ms"], cfg["reduce_scatter_ms"] = gemm_ms, rs_ms
    return cfg


def rowise_ag_gemm_dispatcher(a, b, c, ctx, persistent=False, autotune=False, straggler_option=None):
    from .ag_gemm import ag_gemm_intra_node_op

    return ag_gemm_intra_node_op(a, b, c, ctx)


def local_copy_and_barrier_all(local_rank, rank, num_ranks, local_data, global_data, comm_buf, barrier_ptr, M_per_rank, N,
                               phase, is_internode: bool = False, use_cooperative: bool = True):
    """Stage this rank's rows into its slot of the gathered buffer and meet the peers.  (The fused op does
    not need the barrier - its flags carry a per-call number - this is the stand-alone form.)"""
    from ..utils.compat import nvshmem_barrier_all_on_stream

    global_data[rank * M_per_rank:(rank + 1) * M_per_rank].copy_(local_data)
    nvshmem_barrier_all_on_stream(None)


def cp_engine_producer_all_gather_full_mesh_push_multi_stream(rank, num_ranks, local_tensor, remote_tensor_buffers, one=None,
                                                              M_PER_CHUNK: int = 0, ag_stream_pool=None,
                                                              barrier_buffers=None, **_):
    """The AMD reference spreads the per-peer copies over a pool of streams; SDMA engines are picked per copy by
    the runtime here, so one stream order suffices."""
    stream = ag_stream_pool[0] if ag_stream_pool else None
    return cp_engine_producer_all_gather_intra_node(rank, num_ranks, local_tensor, remote_tensor_buffers, barrier_buffers,
                                                    stream)


def gemm_rs_op(input, weight, ctx, persistent: bool = True, fuse_scatter: bool = False):
    from .gemm_rs import gemm_rs_intra_node_op

    return gemm_rs_intra_node_op(input, weight, ctx)


def _gemm_rs_producer(a, b, c, barrier=None, workspace=None, world_size=1, local_world_size=1, fuse_scatter=False, *_, **__):
    """The GEMM half alone: c = a @ b^T (in the fused op the same engine also scatters and counts tiles)."""
    from .gemm import gemm_nt

    return gemm_nt(a, b, out=c)


gemm_rs_producer_persistent = gemm_rs_producer_non_persistent = _gemm_rs_producer


def matmul_fuse_scatter(a, b, scatter_bufs_ptr, rank, num_ranks, transpose_weight=False):
    """Stand-alone form of the AMD reference's GEMM with scatter epilogue (kernels/amd/gemm_reduce_scatter.py:289):
    computes the product and writes row block ``p`` into ``scatter_bufs_ptr[p]`` slot ``rank``."""
    from .gemm import gemm_nt

    full = gemm_nt(a, b.t().contiguous() if transpose_weight else b)
    m = full.shape[0] // num_ranks
    for p in range(num_ranks):
        scatter_bufs_ptr[p][rank * m:(rank + 1) * m].copy_(full[p * m:(p + 1) * m])
    return full


def ring_reduce_after_scatter(rank, num_ranks, scatter_out: torch.Tensor, stream=None):
    """Sum the ``num_ranks`` row blocks a rank received (kernels/amd/gemm_reduce_scatter.py:264)."""
    M, N = scatter_out.shape
    out = torch.empty(M // num_ranks, N, dtype=scatter_out.dtype, device=scatter_out.device)
    return _rs.ring_reduce(scatter_out, out, rank, num_ranks)


def rowise_ag_scatter_group_gemm_dispatcher(a, b, c, ctx, full_topk_ids):
    out = _moe.ag_group_gemm(a, b, ctx, full_topk_ids)
    if c is not None:
        c.copy_(out)
        return c
    return out


def get_auto_triton_config(M, N, K, N_CHUNKS, dtype):
    """Tile shape of the grouped GEMM for MoE + RS (fixed: the 256 x 256 x 64 engine) and the chunk width."""
    return {"BLOCK_SIZE_M": 256, "BLOCK_SIZE_N": 256, "BLOCK_SIZE_K": 64, "N_PER_CHUNK": N // max(1, N_CHUNKS)}


# ---- GEMM + all-reduce stages (gemm_allreduce.py:464-510) ---------------------------------------------------------
def persistent_gemm_notify(a, b, out, barrier, gemm_config=None, use_tma=False):
    """GEMM alone; ``barrier`` (one int per 16-row block of the output) is raised when the product is complete.
    The fused op raises it per tile from the epilogue; the stand-alone form raises all of it after the GEMM."""
    from .gemm import gemm_nt

    gemm_nt(a, b, out=out)
    barrier.fill_(1)
    return out


def consumer_all_reduce(symm_input, symm_ar_out, ar_out, gemm_barrier=None, multi_st_barrier=None, BLOCK_SIZE_M=16,
                        BLOCK_SIZE_N=64, NUM_COMM_SMS=16, USE_MULTIMEM_ST=False, ctx=None):
    """All-reduce of a finished partial: needs an ``AllReduceContext`` (``ctx``) in place of the reference's raw
    barrier buffers."""
    assert ctx is not None, "pass ctx=create_allreduce_ctx(...)"
    return _ar.all_reduce(symm_input, _ar.AllReduceMethod.OneShot, ctx, ar_out, NUM_COMM_SMS)


# ---- sequence-parallel attention, inter-node names (sp_ag_attention_inter_node.py:56-560) ----------------------
def get_compute_config():
    """(query rows, KV rows per step, waves, stages) of the varlen attention kernel."""
    return (128, 32, 8, 2)


def create_sp_ag_attention_context_inter_node(batch_size, q_head, kv_head, max_seqlen_k, max_q_shard_len, head_dim, input_dtype,
                                              output_dtype, rank, local_rank=None, world_size=1, local_world_size=None,
                                              device=None, **kw):
    from .sp_ag_attention import create_sp_ag_attention_context_intra_node

    assert local_world_size in (None, world_size), "one xGMI node: the inter-node context is the intra-node one"
    return create_sp_ag_attention_context_intra_node(batch_size, q_head, kv_head, max_seqlen_k, max_q_shard_len, head_dim,
                                                     input_dtype, output_dtype, rank, world_size, device, **kw)


def fused_sp_ag_attn_inter_node(ctx, q_shard, k_shard, v_shard, output, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                                rank, world_size, local_world_size=None, is_causal=True, enable_zig_zag=True, **kw):
    from .sp_ag_attention import fused_sp_ag_attn_intra_node

    assert local_world_size in (None, world_size), "one xGMI node"
    return fused_sp_ag_attn_intra_node(ctx, q_shard, k_shard, v_shard, output, cu_seqlens_q, cu_seqlens_k, max_seqlen_q,
                                       max_seqlen_k, rank, world_size, is_causal, enable_zig_zag, **kw)


def cp_engine_producer_kv_all_gather(k_shard, v_shard, k_buffer, v_buffer, k_buffers, v_buffers, cu_seqlens_k, rank, world_size,
                                     signal_buffers=None, signal_target: int = 1, enable_zig_zag: bool = True, stream=None, **_):
    """Copy-engine form of the KV all-gather: every sequence's shard rows go to their place in every peer's full
    K / V buffer (zig-zag: first half-shard at slab ``rank``, second at slab ``2w - 1 - rank``), then the peer's
    flag ``rank`` is raised.  The fused op pushes from a kernel instead (one launch)."""
    from .common_ops import memcpy_async, set_signal

    cu = [int(x) for x in cu_seqlens_k]
    row = k_shard[0].numel() * k_shard.element_size()
    off = 0
    for b in range(len(cu) - 1):
        full = cu[b + 1] - cu[b]
        shard = full // world_size
        pieces = [(0, shard, rank * shard)] if not enable_zig_zag else \
            [(0, shard // 2, rank * (shard // 2)), (shard // 2, shard - shard // 2, (2 * world_size - 1 - rank) * (shard // 2))]
        for src0, n, dst0 in pieces:
            for p in range(world_size):
                for src, bufs in ((k_shard, k_buffers), (v_shard, v_buffers)):
                    d = bufs[p][cu[b] + dst0:cu[b] + dst0 + n]
                    s = src[off + src0:off + src0 + n]
                    if d.is_cuda:
                        memcpy_async(d, s, n * row, stream)
                    else:
                        d.copy_(s)
        off += shard
    if signal_buffers is not None:
        for p in range(world_size):
            set_signal(signal_buffers[p][rank:rank + 1], signal_target, stream)


SPAllGatherAttentionContextInterNode = None  # bound below (import cycle-free)


def _bind_sp_context():
    global SPAllGatherAttentionContextInterNode
    from .sp_ag_attention import SPAllGatherAttentionContextIntraNode

    SPAllGatherAttentionContextInterNode = SPAllGatherAttentionContextIntraNode


_bind_sp_context()


# ---- flash decode: AOT front end and algo-info names (flash_decode.py:103-130, 1095-1150) -------------------------
def get_triton_split_kv_algo_info(*a, **kw):
    from ..tools.aot_kernels import get_split_kv_algo_info

    return get_split_kv_algo_info(*a, **kw)


def get_triton_combine_kv_algo_info(*a, **kw):
    from ..tools.aot_kernels import get_combine_kv_algo_info

    return get_combine_kv_algo_info(*a, **kw)


def get_triton_persistent_algo_info(*a, **kw):
    from ..tools.aot_kernels import get_persistent_algo_info

    return get_persistent_algo_info(*a, **kw)


def gqa_fwd_batch_decode_persistent_aot(stream, q, k_cache, v_cache, workspace, q_lens, kv_lens, block_table, scale, soft_cap=0,
                                        output_split=None, output_combine=None, kv_split=-1):
    """The precompiled form of the persistent decode: every kernel of this package is compiled ahead of time, so
    this is the same launch made on ``stream``."""
    from .flash_decode import gqa_fwd_batch_decode_persistent

    with torch.cuda.stream(stream) if stream is not None and q.is_cuda else _nullcontext():
        return gqa_fwd_batch_decode_persistent(q, k_cache, v_cache, workspace, q_lens, kv_lens, block_table, scale, soft_cap,
                                               output_split, output_combine, kv_split)


def _nullcontext():
    import contextlib

    return contextlib.nullcontext()


# ---- expert-parallel dispatch helpers (ep_a2a.py:337-386) ---------------------------------------------------------
def get_dispatch_send_reqs_for_target_node(token_node_idx: torch.Tensor, traget_node_id: int, index_type=torch.int32):
    """Runs of consecutive tokens that have at least one expert on node ``traget_node_id``: two index vectors of
    length T whose leading entries are the run starts / run ends (exclusive), in order; the rest is padding
    (the remaining positions in ascending order, as a stable descending sort of a 0/1 mask leaves them)."""
    goes = (token_node_idx == traget_node_id).any(dim=1)
    T = goes.shape[0]
    prev = torch.cat((goes.new_zeros(1), goes[:-1]))
    nxt = torch.cat((goes[1:], goes.new_zeros(1)))
    is_start, is_end = goes & ~prev, goes & ~nxt
    pos = torch.arange(T, device=goes.device)

    def packed(mask, shift):
        hit, miss = pos[mask] + shift, None
        m2 = torch.ones(T + 1, dtype=torch.bool, device=goes.device)
        m2[hit] = False
        miss = torch.arange(T + 1, device=goes.device)[m2]
        return torch.cat((hit, miss))[:T].to(index_type)

    return packed(is_start, 0), packed(is_end, 1)


def get_ag_splits_and_recv_offset_for_dispatch(local_splits, full_splits_buf, splits_signal_buf, topk, world_size, experts_per_rank,
                                               cpu_default_val=-1, offset_dtype=torch.int32, num_sm=20):
    """All-gather the per-expert token counts through the symmetric ``full_splits_buf`` [world, E(+)] and derive
    (recv_buf_offset_per_expert [world, experts_per_rank, world], tokens received per rank (CPU), tokens held per
    rank).  Offsets make the rows every source sends to one expert contiguous at the owner."""
    from ..utils.compat import nvshmem_barrier_all_on_stream

    env = get_env()
    E = world_size * experts_per_rank
    for p in range(world_size):
        env.heap.peer_view(full_splits_buf, p)[env.rank, :local_splits.numel()].copy_(local_splits)
    nvshmem_barrier_all_on_stream(None)
    table = full_splits_buf[:, :E].to(torch.int64)                                  # [src, expert]
    t = table.view(world_size, world_size, experts_per_rank).permute(1, 2, 0).contiguous()   # [owner, e_local, src]
    flat = t.view(world_size, -1)
    offs = (torch.cumsum(flat, 1) - flat).view(world_size, experts_per_rank, world_size).to(offset_dtype)
    recv_per_rank = flat.sum(1).to(torch.int32).cpu()
    held = (table.sum(1) // max(1, topk)).to(torch.int32)
    return offs, recv_per_rank, held
