"""Tensor-parallel attention (QKV column-parallel over heads, O row-parallel).

Reference: layers/amd/tp_attn.py:52-311 (layer_norm, rope cache, rope kernel, TP_Attn) plus the
all-reduce / GEMM+AR forward modes of layers/nvidia/tp_attn.py:179-321.  The reference's AMD layer
depends on the external flash_attn package; here prefill and decode attention are the in-tree
gfx950 kernels (ops/flash_attn.py), q/k RMSNorm and RoPE are the fused elementwise kernels.
"""
from __future__ import annotations

import torch

from ..ops import elementwise, flash_decode
from ..ops.ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context
from ..ops.allreduce import AllReduceMethod, all_reduce, create_allreduce_ctx
from ..ops.flash_attn import flash_attn_with_kvcache
from ..ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node
from ..utils.distributed import get_env
from ..ops.gemm import gemm_nt_norm, norm_gemm_supported
from .tp_mlp import _linear, prenorm_generic, shard_local


def layer_norm(hidden_states: torch.Tensor, eps: float, w: torch.Tensor) -> torch.Tensor:
    """RMSNorm over the last dim (the reference keeps the name ``layer_norm``)."""
    return elementwise.rms_norm(hidden_states, w, eps)


def _set_cos_sin_cache(inv_freq: torch.Tensor, max_length: int):
    """(cos, sin) tables [max_length, head_dim / 2] in fp32."""
    t = torch.arange(max_length, device=inv_freq.device, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq.float())
    return freqs.cos().contiguous(), freqs.sin().contiguous()


def apply_rotary_pos_emb(x: torch.Tensor, cos_sin_cache, position_ids: torch.Tensor) -> torch.Tensor:
    """x [bsz, seq, heads, D] (any row stride), position_ids [bsz, seq]; in place, neox layout."""
    bsz, seq, H, D = x.shape
    cos, sin = cos_sin_cache
    x3 = x.view(bsz * seq, H, D) if x.is_contiguous() else x.reshape(bsz * seq, H, D)
    elementwise.apply_rope_(x3, position_ids.reshape(-1), cos, sin)
    return x3.view(bsz, seq, H, D)


class TP_Attn:

    def __init__(self, rank=0, world_size=8, group=None, env=None):
        self.rank = rank
        self.world_size = world_size
        self.group = group
        self.env = env
        self.head_dim = 128
        self.wqkv = None
        self.wo = None
        self.ag_ctx = self.rs_ctx = self.ar_ctx = self.gemm_ar_ctx = None
        self.ar_method = AllReduceMethod.Unknown

    def _device(self):
        try:
            return (self.env or get_env()).device
        except RuntimeError:
            return torch.device("cuda" if torch.cuda.is_available() else "cpu")

    def _init_parameters(self, self_attn, verbose=False):
        dev = self._device()
        self.head_dim = int(getattr(self_attn, "head_dim", self.head_dim))
        self.q_size = self_attn.q_proj.weight.shape[0] // self.world_size
        self.kv_size = self_attn.k_proj.weight.shape[0] // self.world_size
        assert self.kv_size % self.head_dim == 0 and self.kv_size > 0, \
            "KV heads must be divisible by the TP degree"
        wq = shard_local(self_attn.q_proj.weight.detach(), self.world_size, 0, self.rank)
        wk = shard_local(self_attn.k_proj.weight.detach(), self.world_size, 0, self.rank)
        wv = shard_local(self_attn.v_proj.weight.detach(), self.world_size, 0, self.rank)
        self.wqkv = torch.cat((wq, wk, wv), dim=0).to(dev)                       # [(q + 2 kv) / w, H]
        self.wo = shard_local(self_attn.o_proj.weight.detach(), self.world_size, 1, self.rank).to(dev)   # [H, q / w]
        self.ag_N_per_rank = self.wqkv.shape[0]
        self.K = self.wqkv.shape[1]
        self.dtype = self.wqkv.dtype
        if getattr(self_attn, "q_norm", None) is not None:
            self.q_norm_eps = getattr(self_attn.q_norm, "variance_epsilon", getattr(self_attn.q_norm, "eps", 1e-6))
            self.q_norm_w = self_attn.q_norm.weight.detach().to(dev)
        if getattr(self_attn, "k_norm", None) is not None:
            self.k_norm_eps = getattr(self_attn.k_norm, "variance_epsilon", getattr(self_attn.k_norm, "eps", 1e-6))
            self.k_norm_w = self_attn.k_norm.weight.detach().to(dev)
        if verbose:
            print(f"[RANK {self.rank}] Attn qkv {tuple(self.wqkv.shape)} o {tuple(self.wo.shape)}")

    def _init_ctx(self, max_M, ag_intranode_stream=None, BLOCK_M=256, BLOCK_N=256, BLOCK_K=64, stages=2, serial=False,
                  ag_internode_stream=None):
        self.ag_ctx = create_ag_gemm_intra_node_context(
            max_M=max_M, N=self.ag_N_per_rank, K=self.K, rank=self.rank, num_ranks=self.world_size,
            input_dtype=self.dtype, output_dtype=self.dtype, tp_group=self.group, ag_streams=ag_intranode_stream,
            serial=serial, autotune=True, BLOCK_M=BLOCK_M, BLOCK_N=BLOCK_N, BLOCK_K=BLOCK_K, stages=stages,
            M_PER_CHUNK=256, env=self.env)
        self.rs_ctx = create_gemm_rs_intra_node_context(max_M=max_M, N=self.K, rank=self.rank,
                                                        num_ranks=self.world_size, output_dtype=self.dtype,
                                                        tp_group=self.group, fuse_scatter=True, env=self.env)

    def _init_AR_ctx(self, max_M, method: AllReduceMethod = AllReduceMethod.Unknown, dtype=torch.bfloat16):
        self.ar_method = method
        self.ar_ctx = create_allreduce_ctx(workspace_nbytes=max(max_M * self.K * dtype.itemsize, 1 << 16) * 2,
                                           rank=self.rank, world_size=self.world_size,
                                           local_world_size=self.world_size, env=self.env)

    def _init_gemm_ar_ctx(self, max_M, dtype=torch.bfloat16):
        from .gemm_allreduce_layer import GemmARLayer

        N, K = self.wo.shape
        self.gemm_ar_ctx = GemmARLayer(self.group, max_M, N, K, dtype, dtype, self.world_size, persistent=True,
                                       use_ll_kernel=max_M <= 256, copy_to_local=False, rank=self.rank, env=self.env)

    def finalize(self):
        self.ag_ctx = self.rs_ctx = self.ar_ctx = self.gemm_ar_ctx = None

    # ---- shared middle: split, qk-norm, rope, cache update, attention --------------------------
    def _attend(self, qkv: torch.Tensor, position_ids, cos_sin_cache, kv_cache, layer_idx: int) -> torch.Tensor:
        """qkv [bsz, q_len, (q + 2 kv) / w] -> attention output [bsz, q_len, q / w]."""
        bsz, q_len, _ = qkv.shape
        D = self.head_dim
        k_cache, v_cache, kv_offset = kv_cache.update_kv_cache(None, None, layer_idx)
        total = getattr(kv_cache, "total_len", None)
        if (total is not None and kv_offset.dtype == torch.int32 and k_cache.dim() == 4 and k_cache.shape[0] == bsz
                and elementwise.qkv_prep_supported(qkv, k_cache, v_cache, D) and cos_sin_cache[0].is_cuda):
            if q_len == 1 and flash_decode.decode_step_from_qkv_supported(qkv, k_cache, v_cache, D, cos_sin_cache[0]):
                # a decode step: the attention kernel norms, rotates and appends on its own - one launch in all
                out = flash_decode.decode_step_from_qkv(
                    qkv, getattr(self, "q_norm_w", None), getattr(self, "k_norm_w", None),
                    getattr(self, "q_norm_eps", 0.0), getattr(self, "k_norm_eps", 0.0), position_ids, cos_sin_cache[0],
                    cos_sin_cache[1], k_cache, v_cache, kv_offset, total, self.q_size // D, self.kv_size // D,
                    seq_table=getattr(kv_cache, "seq_table", None))
                return out.reshape(bsz, q_len, -1)
            # one launch: per-head norms, rotation, cache append, attended length (ops/elementwise.py: qkv_prep)
            q = elementwise.qkv_prep(qkv, getattr(self, "q_norm_w", None), getattr(self, "k_norm_w", None),
                                     getattr(self, "q_norm_eps", 0.0), getattr(self, "k_norm_eps", 0.0), position_ids,
                                     cos_sin_cache[0], cos_sin_cache[1], k_cache, v_cache, kv_offset, total,
                                     self.q_size // D, self.kv_size // D, D)
            out = flash_attn_with_kvcache(q=q, k_cache=k_cache, v_cache=v_cache, cache_seqlens=total, causal=True,
                                          seq_table=getattr(kv_cache, "seq_table", None))
            return out.reshape(bsz, q_len, -1)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        v = v.reshape(bsz, q_len, -1, D)
        if hasattr(self, "q_norm_eps"):
            q = layer_norm(q.reshape(-1, D), self.q_norm_eps, self.q_norm_w)
        if hasattr(self, "k_norm_eps"):
            k = layer_norm(k.reshape(-1, D), self.k_norm_eps, self.k_norm_w)
        q = apply_rotary_pos_emb(q.reshape(bsz, q_len, -1, D), cos_sin_cache, position_ids)
        k = apply_rotary_pos_emb(k.reshape(bsz, q_len, -1, D), cos_sin_cache, position_ids)
        k_cache, v_cache, kv_offset = kv_cache.update_kv_cache(k, v, layer_idx)
        out = flash_attn_with_kvcache(q=q, k_cache=k_cache, v_cache=v_cache, k=k, v=v, cache_seqlens=kv_offset,
                                      causal=True)
        return out.reshape(bsz, q_len, -1)

    def _all_reduce_torch(self, out):
        if self.world_size > 1 and torch.distributed.is_initialized():
            torch.distributed.all_reduce(out, torch.distributed.ReduceOp.SUM, group=self.group)
        return out

    # ---- forward modes ---------------------------------------------------------------------------
    @torch.inference_mode()
    def torch_fwd(self, x, position_ids, cos_sin_cache, kv_cache, layer_idx: int):
        """x replicated [bsz, q_len, H]; output all-reduced."""
        qkv = torch.nn.functional.linear(x, self.wqkv)
        out = self._attend(qkv, position_ids, cos_sin_cache, kv_cache, layer_idx)
        out = torch.nn.functional.linear(out, self.wo)
        return self._all_reduce_torch(out)

    @torch.inference_mode()
    def dist_triton_fwd(self, x, position_ids, cos_sin_cache, kv_cache, layer_idx: int):
        """x batch-sharded [bsz / w, q_len, H] -> same shape (AG+GEMM ... GEMM+RS)."""
        bsz, q_len, d = x.shape
        w = self.world_size
        qkv = ag_gemm_intra_node(x.reshape(-1, d), self.wqkv, transe_b=False, ctx=self.ag_ctx).view(bsz * w, q_len, -1)
        out = self._attend(qkv, position_ids, cos_sin_cache, kv_cache, layer_idx)
        out = gemm_rs_intra_node(out.reshape(bsz * w * q_len, -1), self.wo, self.rs_ctx)
        return out.view(bsz, q_len, -1)

    @torch.inference_mode()
    def dist_triton_AR_fwd(self, x, position_ids, cos_sin_cache, kv_cache, layer_idx: int, qkv=None):
        """``qkv``: the projection when the caller has computed it already (``fwd_prenorm``; x then only gives the shape)."""
        qkv = _linear(x, self.wqkv) if qkv is None else qkv
        out = self._attend(qkv, position_ids, cos_sin_cache, kv_cache, layer_idx)
        out = _linear(out, self.wo)
        if self.world_size > 1:
            assert self.ar_ctx is not None, "AllReduce context is not initialized."
            out = all_reduce(out.contiguous(), output=torch.empty_like(out), method=self.ar_method, ctx=self.ar_ctx)
        return out.view_as(x)

    @torch.inference_mode()
    def dist_triton_gemm_ar_fwd(self, x, position_ids, cos_sin_cache, kv_cache, layer_idx: int, qkv=None):
        bsz, q_len, _ = x.shape
        qkv = _linear(x, self.wqkv) if qkv is None else qkv
        out = self._attend(qkv, position_ids, cos_sin_cache, kv_cache, layer_idx)
        assert self.gemm_ar_ctx is not None, "GemmAR context is not initialized."
        out = self.gemm_ar_ctx.forward(out.reshape(bsz * q_len, -1), self.wo)
        return out.view(bsz, q_len, -1)

    def fwd(self, x, position_ids, cos_sin_cache, kv_cache, layer_idx: int):
        raise NotImplementedError("Please use torch_fwd or dist_triton_fwd instead.")

    @torch.inference_mode()
    def fwd_prenorm(self, x, residual, norm_w, eps, position_ids, cos_sin_cache, kv_cache, layer_idx: int):
        """RMSNorm(x [+ residual]) -> this block: (block output, x + residual).  Decode-sized inputs of the modes
        whose QKV projection is a local GEMM run the norm inside that GEMM's launch (``gemm_nt_norm``)."""
        fn = getattr(self.fwd, "__func__", None)
        # (on ONE rank the AG + GEMM / GEMM + RS mode is local GEMMs too: it takes the all-reduce form, minus the reduce)
        solo = fn is TP_Attn.dist_triton_fwd and self.world_size == 1
        if fn in (TP_Attn.dist_triton_AR_fwd, TP_Attn.dist_triton_gemm_ar_fwd) or solo:
            x2 = x.reshape(-1, x.shape[-1])
            r2 = residual.reshape(x2.shape) if residual is not None else None
            if norm_gemm_supported(x2, self.wqkv, norm_w, r2):
                qkv, hidden = gemm_nt_norm(x2, norm_w, eps, self.wqkv, residual=r2)
                shape = residual.shape if residual is not None else x.shape
                fwd = self.dist_triton_AR_fwd if solo else self.fwd
                out = fwd(x.reshape(shape), position_ids, cos_sin_cache, kv_cache, layer_idx, qkv=qkv.view(*shape[:-1], -1))
                return out, hidden.view(shape)
        return prenorm_generic(self, x, residual, norm_w, eps, position_ids, cos_sin_cache, kv_cache, layer_idx)
