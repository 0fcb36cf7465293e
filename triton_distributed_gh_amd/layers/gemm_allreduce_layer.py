"""GEMM + all-reduce layer wrapper (reference: layers/nvidia/gemm_allreduce_layer.py:33-116).

The reference picks between a low-latency and a throughput kernel configuration; so does ``gemm_allreduce_op``:
up to 256 rows the low-latency kernel (split-K partials pushed to every peer with the flag in the data, one
launch), above that the persistent 256-row tile kernel whose 32-row reduce slabs are queued a tile group behind
the GEMM tiles (csrc/ops/gemm_ar.hip).  The choice is made per call from the row count (``TDIST_GEMM_AR_LL_MAX_M``); ``use_ll_kernel`` is accepted for the reference's signature and recorded."""
from __future__ import annotations

import torch

from ..ops.gemm_ar import create_gemm_ar_context, gemm_allreduce_op
from ..utils.distributed import get_env


class GemmARLayer(torch.nn.Module):

    def __init__(self, tp_group, max_M, N, K, input_dtype, output_dtype, local_world_size, persistent=True,
                 use_ll_kernel=False, copy_to_local=True, NUM_COMM_SMS=16, BLOCK_M=256, BLOCK_N=256, BLOCK_K=64,
                 stages=2, rank=None, env=None):
        super().__init__()
        env = env or get_env()
        self.env = env
        self.tp_group = tp_group
        self.rank = env.rank if rank is None else rank
        self.world_size = env.world_size
        self.max_M, self.N, self.K = max_M, N, K
        self.input_dtype, self.output_dtype = input_dtype, output_dtype
        self.use_ll_kernel = use_ll_kernel
        self.copy_to_local = copy_to_local
        self.persistent = persistent
        num_ctas = 0        # the kernels size their grids themselves (ranks sharing one GPU split its CUs)
        self.ctx = create_gemm_ar_context(None, self.rank, self.world_size, local_world_size, max_M, N, output_dtype,
                                          NUM_COMM_SMS=NUM_COMM_SMS, num_ctas=num_ctas, env=env)

    def forward(self, x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """x [M, K_local], weight [N, K_local] -> all-reduced [M, N]."""
        return gemm_allreduce_op(self.ctx, x, weight, copy_to_local=self.copy_to_local)

    def finalize(self):
        self.ctx = None
