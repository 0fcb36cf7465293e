"""Entry points under the names of the reference's NVIDIA kernel library
(reference: python/triton_dist/kernels/nvidia/__init__.py:25-87) mapped onto the MI355X ops.

``create_ag_gemm_context`` / ``ag_gemm`` (allgather_gemm.py:489-575) and ``create_gemm_rs_context`` /
``gemm_rs`` (gemm_reduce_scatter.py:70-82, 569-583) take the NVIDIA argument lists; tile-size,
stage and SM-partition arguments are accepted and ignored (the gfx950 kernels have one tuned
256x256x64 configuration and share CUs dynamically)."""
from __future__ import annotations

import torch

from ..utils.distributed import get_env
from .ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context
from .allgather import all_gather_full_mesh
from .allreduce import _run_straggler
from .gemm import gemm_nt
from .gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node


def create_ag_gemm_context(tensor_A, tensor_B, rank, num_ranks, max_M, num_local_ranks=8, BLOCK_M=256, BLOCK_N=256,
                           BLOCK_K=64, stages=2, ag_intranode_stream=None, ag_internode_stream=None,
                           for_correctness=False, env=None):
    assert tensor_A.shape[1] == tensor_B.shape[1], "B is [N_per_rank, K]"
    streams = ag_intranode_stream if isinstance(ag_intranode_stream, (list, tuple)) or ag_intranode_stream is None \
        else [ag_intranode_stream]
    ctx = create_ag_gemm_intra_node_context(max_M, tensor_B.shape[0], tensor_A.shape[1], tensor_A.dtype, tensor_A.dtype,
                                            rank, num_ranks, ag_streams=streams, env=env)
    ctx.for_correctness = for_correctness
    return ctx


def ag_gemm(a, b, ctx, persistent=True, autotune=False, straggler_option=None, use_cooperative=True):
    """a [M_per_rank, K], b [N_per_rank, K] -> [M, N_per_rank]."""
    _run_straggler(straggler_option, ctx.rank)
    return ag_gemm_intra_node(a, b, False, ctx)


def gemm_persistent(a, b, c=None, **_):
    return gemm_nt(a, b, out=c)


gemm_non_persistent = gemm_persistent


def create_gemm_rs_context(max_M, N, rank, world_size, local_world_size, output_dtype, rs_stream=None, BLOCK_M=256,
                           BLOCK_N=256, BLOCK_K=64, GROUP_M=8, stages=2, env=None):
    assert world_size == local_world_size, "single-node only"
    ctx = create_gemm_rs_intra_node_context(max_M, N, output_dtype, rank, world_size, env=env)
    ctx.rs_stream = rs_stream
    return ctx


def gemm_rs(a, b, ctx, persistent=True, fuse_scatter=False):
    """a [M, K_local], b [N, K_local] -> [M // world_size, N].  The scatter is always fused into the
    GEMM epilogue and the reduction pipelined behind it (one persistent launch)."""
    return gemm_rs_intra_node(a, b, ctx)


def cp_engine_producer_all_gather_intra_node(rank, num_ranks, local_tensor, remote_buffers, barrier_buffers=None,
                                             intranode_ag_stream=None, for_correctness=False, all_gather_method=None,
                                             env=None):
    """Full-mesh copy-engine all-gather of ``local_tensor`` into slot ``rank`` of every rank's
    ``remote_buffers[r]`` (list of per-rank views of one symmetric [num_ranks * m, ...] buffer);
    flags (if given) are set to 1 per source rank, in stream order after the data."""
    from .common_ops import memcpy_async, set_signal

    env = env or get_env()
    m = local_tensor.shape[0]
    nbytes = local_tensor.numel() * local_tensor.element_size()
    for i in range(num_ranks):
        peer = (rank + i) % num_ranks
        dst = remote_buffers[peer][rank * m:(rank + 1) * m]
        if env.backend == "host":
            dst.copy_(local_tensor)
        else:
            memcpy_async(dst, local_tensor, nbytes, intranode_ag_stream)
        if barrier_buffers is not None:
            set_signal(barrier_buffers[peer][rank:rank + 1], 1, intranode_ag_stream)


cp_engine_producer_all_gather_inter_node = cp_engine_producer_all_gather_intra_node
__all__ = ["create_ag_gemm_context", "ag_gemm", "gemm_persistent", "gemm_non_persistent", "create_gemm_rs_context",
           "gemm_rs", "cp_engine_producer_all_gather_intra_node", "cp_engine_producer_all_gather_inter_node",
           "all_gather_full_mesh"]
_ = torch
