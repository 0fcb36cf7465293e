"""Contiguous per-sequence KV cache, heads sharded over the tensor-parallel ranks.

Layout ``[layers, batch, max_length, kv_heads / world_size, head_dim]`` with one running offset per
sequence (reference: python/triton_dist/models/kv_cache.py:28-66).  288 GB of HBM per MI355X holds
e.g. Qwen3-32B at batch 4096 x 4 K context per rank without paging, so the dense engine keeps this
flat form; the paged form used by the flash-decode kernels is ``PagedKVCache``."""
from __future__ import annotations

import torch


def _default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class KV_Cache:

    def __init__(self, num_layers: int = 32, batch_size: int = 1, max_length: int = 32 * 1024, kv_heads: int = 8,
                 head_dim: int = 128, dtype=torch.bfloat16, world_size: int = 8, device=None) -> None:
        self.num_layers, self.batch_size, self.max_length = num_layers, batch_size, max_length
        self.kv_heads, self.head_dim, self.dtype, self.world_size = kv_heads, head_dim, dtype, world_size
        self.device = torch.device(device) if device is not None else _default_device()
        assert kv_heads % world_size == 0, "kv heads must divide over the TP ranks"
        shape = (num_layers, batch_size, max_length, kv_heads // world_size, head_dim)
        self.k_cache = torch.zeros(shape, device=self.device, dtype=dtype)
        self.v_cache = torch.zeros(shape, device=self.device, dtype=dtype)
        self.kv_offset = torch.zeros(batch_size, dtype=torch.int32, device=self.device)
        # scratch of the fused norm + rope + append kernel (lengths including the tokens of this
        # step) and the one-page-per-sequence table of the decode kernel: created once here so that a
        # decode step launches no index arithmetic
        self.total_len = torch.zeros(batch_size, dtype=torch.int32, device=self.device)
        self.seq_table = torch.arange(batch_size, dtype=torch.int32, device=self.device).view(batch_size, 1)

    def update_kv_cache(self, new_k_cache: torch.Tensor, new_v_cache: torch.Tensor, layer_idx: int):
        """Returns the layer's cache views and offsets; the attention op appends k/v itself."""
        return self.k_cache[layer_idx], self.v_cache[layer_idx], self.kv_offset

    def rand_fill_kv_cache(self, offset: int):
        shape = self.k_cache[:, :, :offset].size()
        self.k_cache[:, :, :offset].copy_(torch.rand(shape, device=self.device, dtype=self.dtype) / 10)
        self.v_cache[:, :, :offset].copy_(torch.rand(shape, device=self.device, dtype=self.dtype) / 10)

    def inc_offset(self, n: int = 1):
        self.kv_offset += n

    def clear(self):
        self.kv_offset.zero_()

    def get_kv_len(self):
        return self.kv_offset


class PagedKVCache:
    """Page-table KV cache: ``[layers, pages, page_size, kv_heads / world_size, head_dim]`` plus a
    ``block_table [batch, max_pages]`` (reference: mega_triton_kernel/models/paged_kv_cache.py)."""

    def __init__(self, num_layers, batch_size, max_length, kv_heads, head_dim, page_size=64, dtype=torch.bfloat16,
                 world_size=1, device=None):
        self.device = torch.device(device) if device is not None else _default_device()
        self.num_layers, self.batch_size, self.max_length, self.page_size = num_layers, batch_size, max_length, page_size
        self.kv_heads, self.head_dim, self.dtype, self.world_size = kv_heads, head_dim, dtype, world_size
        self.pages_per_seq = (max_length + page_size - 1) // page_size
        pages = batch_size * self.pages_per_seq
        shape = (num_layers, pages, page_size, kv_heads // world_size, head_dim)
        self.k_cache = torch.zeros(shape, device=self.device, dtype=dtype)
        self.v_cache = torch.zeros(shape, device=self.device, dtype=dtype)
        self.block_table = torch.arange(pages, dtype=torch.int32, device=self.device).view(batch_size, -1)
        self.kv_offset = torch.zeros(batch_size, dtype=torch.int32, device=self.device)

    def slots(self, positions: torch.Tensor) -> torch.Tensor:
        """positions [batch, n] -> flat row index into ``cache[layer].view(-1, heads, dim)``."""
        page = torch.gather(self.block_table.long(), 1, (positions // self.page_size).long())
        return (page * self.page_size + positions % self.page_size).reshape(-1)

    def update_kv_cache(self, new_k, new_v, layer_idx):
        return self.k_cache[layer_idx], self.v_cache[layer_idx], self.kv_offset

    def rand_fill_kv_cache(self, offset: int):
        self.k_cache.copy_(torch.rand_like(self.k_cache) / 10)
        self.v_cache.copy_(torch.rand_like(self.v_cache) / 10)
        self.kv_offset.fill_(offset)

    def inc_offset(self, n: int = 1):
        self.kv_offset += n

    def clear(self):
        self.kv_offset.zero_()

    def get_kv_len(self):
        return self.kv_offset
