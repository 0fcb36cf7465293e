"""Sequence-parallel GQA decode attention: KV cache sharded over ranks along the sequence.

Reference: layers/nvidia/sp_flash_decode_layer.py:44-185.  Every rank runs split-KV flash decode on
its KV shard (one launch; the last workgroup merges the splits and emits the LSE), the
[batch, q_heads, D + 1] fp32 partials are all-gathered with the flag-in-data low-latency protocol
and merged by the inter-rank combine kernel, skipping ranks with no KV for a sequence."""
from __future__ import annotations

import torch

from ..ops.flash_decode import gqa_fwd_batch_decode_inter_rank_combine, gqa_fwd_batch_decode_intra_rank
from ..utils.distributed import get_env
from .low_latency_allgather_layer import AllGatherLayer


class SpGQAFlashDecodeAttention(torch.nn.Module):

    def __init__(self, rank, node, num_ranks, num_nodes, num_q_heads, num_kv_heads, q_head_dim, v_head_dim, page_size=1,
                 scale=1, soft_cap=0, max_allowed_batch=1, thrink_buffer_threshold=500, stages=4, env=None):
        super().__init__()
        self.env = env or get_env()
        self.rank, self.node, self.num_ranks, self.num_nodes = rank, node, num_ranks, num_nodes
        self.num_q_heads, self.num_kv_heads = num_q_heads, num_kv_heads
        self.q_head_dim, self.v_head_dim = q_head_dim, v_head_dim
        self.page_size, self.scale, self.soft_cap = page_size, scale, soft_cap
        self.kv_split = -1                      # auto: sized for the CU count
        self.max_allowed_batch = max_allowed_batch
        self.stages = stages
        self.workspace = None
        self.bytes_per_rank = max_allowed_batch * num_q_heads * (v_head_dim + 1) * 4
        self.ag_layer = AllGatherLayer(num_nodes, num_ranks, rank, max_buffer_size=self.bytes_per_rank, stages=stages,
                                       env=self.env)
        self.ag_buffer = self.env.heap.create_tensor([stages, num_ranks * self.bytes_per_rank], torch.int8)
        if self.env.backend == "hip":
            torch.cuda.synchronize()
        if self.env.cpu_group is not None:
            torch.distributed.barrier(group=self.env.cpu_group)

    def finalize(self):
        self.ag_layer.finalize()

    def forward(self, q, k_cache, v_cache, global_kv_lens, block_table):
        """q [batch, q_heads, D] (same on every rank); k/v_cache: this rank's pages
        [pages, page_size, kv_heads, D]; global_kv_lens int32 [num_ranks, batch]; block_table
        [batch, max_pages] for this rank's shard.  Returns [batch, q_heads, D]."""
        batch = q.shape[0]
        D = self.v_head_dim
        assert global_kv_lens.shape == (self.num_ranks, batch)
        assert batch <= self.max_allowed_batch, f"only {self.max_allowed_batch} queries per decode step"
        part = gqa_fwd_batch_decode_intra_rank(q, k_cache, v_cache, self.workspace, [1] * batch,
                                               global_kv_lens[self.rank].contiguous(), block_table, self.scale,
                                               soft_cap=self.soft_cap, kv_split=self.kv_split)
        if self.num_ranks == 1:
            return part[..., :D].to(q.dtype)
        nbytes_per_rank = part.numel() * 4
        stage = self.ag_buffer[self.ag_layer.signal_target % self.stages][: self.num_ranks * nbytes_per_rank]
        stage = stage.view(self.num_ranks, nbytes_per_rank)
        stage[self.rank].copy_(part.view(-1).view(torch.int8))
        self.ag_layer.forward_push_2d_ll(stage)
        allp = stage.view(torch.float32).view(self.num_ranks, batch, self.num_q_heads, D + 1)
        return gqa_fwd_batch_decode_inter_rank_combine(allp, global_kv_lens.contiguous(), dtype=q.dtype)
