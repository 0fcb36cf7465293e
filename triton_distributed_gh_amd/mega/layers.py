"""Layer builders of the mega-kernel runtime: record one tensor-parallel attention / MLP block as tasks.

Reference: python/triton_dist/mega_triton_kernel/models/layers/tp_attn.py:45-140 (``TPAttnBuilder``) and
tp_mlp.py:45-110 (``TPMLPBuilder``): ``_init_parameters`` shards a module's weights, ``build_fwd`` records
the block's ops on a ``ModelBuilder`` and returns the output tensor.  Differences: the builders also accept
the already sharded tensors of this package's ``TP_Attn`` / ``TP_MLP`` layers (``from_layer``), the SiLU * up
activation rides in the epilogue of the first projection (one task type instead of two), and the residual add
is folded into the following norm task by the caller (``Qwen3Model``).
"""
from __future__ import annotations

from typing import Optional

import torch

from ..layers.tp_mlp import shard_local
from .tasks import FlashAttnConfig, LinearConfig


def prepare_cos_sin_cache(head_dim: int, max_position_embeddings: int, rope_theta: float, device=None):
    """fp32 (cos, sin) tables [positions, head_dim / 2] for the rope task (reference:
    python/triton_dist/mega_triton_kernel/models/utils.py:28)."""
    inv_freq = 1.0 / (rope_theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32, device=device) / head_dim))
    t = torch.arange(max_position_embeddings, dtype=torch.float32, device=device)
    freqs = torch.outer(t, inv_freq)
    return freqs.cos().contiguous(), freqs.sin().contiguous()


class TPMLPBuilder:
    """Column-parallel gate / up projection, row-parallel down projection, all-reduce at world size > 1."""

    def __init__(self, builder, rank: int = 0, world_size: int = 8, group=None):
        self._builder, self.rank, self.world_size, self.group = builder, rank, world_size, group
        self.gate_up_proj: Optional[torch.Tensor] = None      # [2 I / w, H]: gate rows, then up rows
        self.down_proj: Optional[torch.Tensor] = None          # [H, I / w]

    def _init_parameters(self, mlp, verbose: bool = False):
        """``mlp``: a module with ``gate_proj`` / ``up_proj`` / ``down_proj`` linear layers (no bias)."""
        assert getattr(mlp.gate_proj, "bias", None) is None, "bias is not supported"
        dev = self._builder.device
        gate = shard_local(mlp.gate_proj.weight.detach(), self.world_size, 0, self.rank)
        up = shard_local(mlp.up_proj.weight.detach(), self.world_size, 0, self.rank)
        self.gate_up_proj = torch.cat((gate, up), dim=0).to(dev).contiguous()
        self.down_proj = shard_local(mlp.down_proj.weight.detach(), self.world_size, 1, self.rank).to(dev).contiguous()
        if verbose:
            print(f"[RANK {self.rank}] mega MLP: gate_up {tuple(self.gate_up_proj.shape)} down {tuple(self.down_proj.shape)}")
        return self

    def from_layer(self, tp_mlp):
        """Take the shards of a ``layers.TP_MLP`` (same layout)."""
        self.gate_up_proj, self.down_proj = tp_mlp.gate_up_proj, tp_mlp.down_proj
        return self

    def build_fwd(self, x: torch.Tensor, act_out: Optional[torch.Tensor] = None, fc2_out: Optional[torch.Tensor] = None,
                  ar_out: Optional[torch.Tensor] = None, name: str = "mlp",
                  fc1_config: Optional[LinearConfig] = None, fc2_config: Optional[LinearConfig] = None) -> torch.Tensor:
        """x [tokens, H] (or [bsz, seq, H]) bf16 -> [.., H]: the all-reduced block output (NOT yet added to the
        residual stream)."""
        b = self._builder
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        T, H = x2.shape
        inter = self.gate_up_proj.shape[0] // 2
        new = lambda *s: torch.zeros(*s, dtype=x.dtype, device=x.device)  # noqa: E731
        act_out = new(T, inter) if act_out is None else act_out
        if self.world_size > 1:
            fc2_out = b.create_symm_tensor((T, H), x.dtype) if fc2_out is None else fc2_out
            ar_out = new(T, H) if ar_out is None else ar_out
        else:
            fc2_out = new(T, H) if fc2_out is None else fc2_out
        b.make_fc1(x2, self.gate_up_proj, act_out, name=f"{name}.fc1", config=fc1_config)
        b.make_fc2(act_out, self.down_proj, fc2_out, name=f"{name}.fc2", config=fc2_config)
        if self.world_size > 1:
            b.make_allreduce(fc2_out, ar_out, name=f"{name}.ar")
            return ar_out.view(shape)
        return fc2_out.view(shape)


class TPAttnBuilder:
    """Fused QKV projection, per-head q / k RMSNorm + rope + paged KV-cache append, split-KV decode attention
    with combine, output projection, all-reduce at world size > 1."""

    def __init__(self, builder, layer_idx: int, head_dim: int = 128, rank: int = 0, world_size: int = 8, group=None):
        self._builder, self.layer_idx, self.head_dim = builder, layer_idx, head_dim
        self.rank, self.world_size, self.group = rank, world_size, group
        self.wqkv = self.wo = self.q_norm_w = self.k_norm_w = None
        self.q_norm_eps = self.k_norm_eps = 1e-6
        self.num_q_heads = self.num_kv_heads = 0

    def _init_parameters(self, self_attn, verbose: bool = False):
        """``self_attn``: a module with ``q_proj`` / ``k_proj`` / ``v_proj`` / ``o_proj`` (and optionally
        ``q_norm`` / ``k_norm``) in the HuggingFace layout."""
        dev, w, r = self._builder.device, self.world_size, self.rank
        q = shard_local(self_attn.q_proj.weight.detach(), w, 0, r)
        k = shard_local(self_attn.k_proj.weight.detach(), w, 0, r)
        v = shard_local(self_attn.v_proj.weight.detach(), w, 0, r)
        self.wqkv = torch.cat((q, k, v), dim=0).to(dev).contiguous()
        self.wo = shard_local(self_attn.o_proj.weight.detach(), w, 1, r).to(dev).contiguous()
        self.num_q_heads, self.num_kv_heads = q.shape[0] // self.head_dim, k.shape[0] // self.head_dim
        if getattr(self_attn, "q_norm", None) is not None:
            self.q_norm_w = self_attn.q_norm.weight.detach().to(dev)
            self.k_norm_w = self_attn.k_norm.weight.detach().to(dev)
            self.q_norm_eps = float(getattr(self_attn.q_norm, "variance_epsilon", getattr(self_attn.q_norm, "eps", 1e-6)))
        if verbose:
            print(f"[RANK {self.rank}] mega attention {self.layer_idx}: wqkv {tuple(self.wqkv.shape)} wo {tuple(self.wo.shape)}")
        return self

    def from_layer(self, tp_attn, num_q_heads: int, num_kv_heads: int):
        """Take the shards of a ``layers.TP_Attn``; head counts are per rank."""
        self.wqkv, self.wo = tp_attn.wqkv, tp_attn.wo
        self.q_norm_w, self.k_norm_w = getattr(tp_attn, "q_norm_w", None), getattr(tp_attn, "k_norm_w", None)
        self.q_norm_eps = getattr(tp_attn, "q_norm_eps", 1e-6)
        self.num_q_heads, self.num_kv_heads = num_q_heads, num_kv_heads
        return self

    def build_fwd(self, x: torch.Tensor, cos_cache: torch.Tensor, sin_cache: torch.Tensor, kv_cache,
                  qkv: Optional[torch.Tensor] = None, q: Optional[torch.Tensor] = None,
                  attn: Optional[torch.Tensor] = None, o_out: Optional[torch.Tensor] = None,
                  ar_out: Optional[torch.Tensor] = None, name: Optional[str] = None,
                  attn_config: Optional[FlashAttnConfig] = None) -> torch.Tensor:
        """x [tokens, H] (normed hidden states, one token per sequence) -> [tokens, H]: the all-reduced block
        output.  ``kv_cache`` is a ``models.kv_cache.PagedKVCache`` whose ``kv_offset`` holds every
        sequence's position; the step's K / V are appended to it by the rope task."""
        b, li, D = self._builder, self.layer_idx, self.head_dim
        name = name or f"l{li}"
        B, H = x.shape
        Hq, Hkv = self.num_q_heads, self.num_kv_heads
        new = lambda *s: torch.zeros(*s, dtype=x.dtype, device=x.device)  # noqa: E731
        qkv = new(B, (Hq + 2 * Hkv) * D) if qkv is None else qkv
        q = new(B, Hq, D) if q is None else q
        attn = new(B, Hq, D) if attn is None else attn
        if self.world_size > 1:
            o_out = b.create_symm_tensor((B, H), x.dtype) if o_out is None else o_out
            ar_out = new(B, H) if ar_out is None else ar_out
        else:
            o_out = new(B, H) if o_out is None else o_out
        b.make_qkv_proj(x, self.wqkv, qkv, name=f"{name}.qkv")
        b.make_qk_norm_rope_update_kvcache(qkv, self.q_norm_w, self.k_norm_w, cos_cache, sin_cache, kv_cache.kv_offset,
                                           kv_cache.k_cache[li], kv_cache.v_cache[li], kv_cache.block_table, q,
                                           self.q_norm_eps, name=f"{name}.qk_norm_rope_kv")
        b.make_attn(q, kv_cache.k_cache[li], kv_cache.v_cache[li], kv_cache.block_table, kv_cache.kv_offset, attn,
                    num_splits=getattr(attn_config, "NUM_KV_SPLITS", 0) or 0, name=f"{name}.attn")
        b.make_o_proj(attn.view(B, Hq * D), self.wo, o_out, name=f"{name}.o_proj")
        if self.world_size > 1:
            b.make_allreduce(o_out, ar_out, name=f"{name}.attn_ar")
            return ar_out
        return o_out
