"""Tensor-parallel decoder stack shared by the dense and MoE model families.

Reference: python/triton_dist/models/dense.py:52-241 (``DenseLLMLayer`` / ``DenseLLM``) and
models/qwen_moe.py:50-206 (``Qwen3MoELayer`` / ``Qwen3MoE``) - two near-identical classes there, one
parametrised stack here.  Differences that matter on MI355X:

* every residual add is folded into the following RMSNorm kernel (one pass over the activations
  instead of two);
* weights are generated / loaded one layer at a time straight onto the GPU (models/weights.py);
* the MoE family has a working ``triton_dist`` mode (the reference raises NotImplemented on AMD,
  models/qwen_moe.py:164-165).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as F

from ..layers.tp_attn import TP_Attn, _set_cos_sin_cache
from ..layers.tp_mlp import TP_MLP, _linear
from ..layers.tp_moe import TP_MoE
from ..ops import elementwise
from ..ops.allreduce import AllReduceMethod
from ..utils.distributed import get_env, is_initialized
from .config import ModelConfig, load_arch_config
from .kv_cache import KV_Cache
from .utils import init_model_cpu

_MODES = {
    "torch": ("torch_fwd", "torch_fwd"),
    "triton_dist": ("dist_triton_fwd", "dist_triton_fwd"),
    "triton_dist_AR": ("dist_triton_AR_fwd", "dist_triton_AR_fwd"),
    "triton_dist_gemm_ar": ("dist_triton_gemm_ar_fwd", "dist_triton_gemm_ar_fwd"),
}


def _eps(norm) -> float:
    return float(getattr(norm, "variance_epsilon", getattr(norm, "eps", 1e-6)))


class DecoderLayer:
    """One pre-norm decoder block: attention + (dense | MoE) feed-forward."""

    def __init__(self, layer_idx: int, group, moe: bool = False, env=None) -> None:
        self.attn: Optional[TP_Attn] = None
        self.mlp = None
        self.input_norm_eps = self.post_norm_eps = None
        self.input_norm_w = self.post_norm_w = None
        self.layer_idx, self.group, self.moe, self.env = layer_idx, group, moe, env

    def init_parameters(self, hf_layer, rank: int, world_size: int):
        dev = self.env.device if self.env is not None else (get_env().device if is_initialized() else
                                                            hf_layer.input_layernorm.weight.device)
        self.mlp = (TP_MoE if self.moe else TP_MLP)(rank=rank, world_size=world_size, group=self.group, env=self.env)
        self.mlp._init_parameters(hf_layer.mlp)
        self.attn = TP_Attn(rank=rank, world_size=world_size, group=self.group, env=self.env)
        self.attn._init_parameters(hf_layer.self_attn)
        self.input_norm_eps = _eps(hf_layer.input_layernorm)
        self.input_norm_w = hf_layer.input_layernorm.weight.detach().to(dev)
        self.post_norm_eps = _eps(hf_layer.post_attention_layernorm)
        self.post_norm_w = hf_layer.post_attention_layernorm.weight.detach().to(dev)

    def set_fwd(self, mode: str = "torch"):
        if mode not in _MODES or (self.moe and mode not in ("torch", "triton_dist")):
            raise ValueError(f"Unsupported mode: {mode}, choose from {list(_MODES)}")
        a, m = _MODES[mode]
        self.attn.fwd = getattr(self.attn, a)
        self.mlp.fwd = getattr(self.mlp, m)

    @torch.inference_mode()
    def fwd_fused(self, hidden, delta, position_ids, cos_sin_cache, kv_cache):
        """``hidden + delta`` is the block input (``delta`` None for the first block).  Returns the
        new (hidden, delta) pair so the next block's norm can absorb the pending add."""
        # (the norms travel with the blocks: decode-sized inputs fold them into the first projection's launch)
        if delta is None:
            a, hidden = self.attn.fwd_prenorm(hidden, None, self.input_norm_w, self.input_norm_eps, position_ids,
                                              cos_sin_cache, kv_cache, self.layer_idx)
        else:
            a, hidden = self.attn.fwd_prenorm(delta, hidden, self.input_norm_w, self.input_norm_eps, position_ids,
                                              cos_sin_cache, kv_cache, self.layer_idx)
        if hasattr(self.mlp, "fwd_prenorm"):
            m, hidden = self.mlp.fwd_prenorm(a, hidden, self.post_norm_w, self.post_norm_eps)
        else:
            x, hidden = elementwise.rms_norm(a, self.post_norm_w, self.post_norm_eps, residual=hidden)
            m = self.mlp.fwd(x)
        return hidden, m

    @torch.inference_mode()
    def fwd(self, hidden_states, position_ids, cos_sin_cache, kv_cache):
        hidden, delta = self.fwd_fused(hidden_states, None, position_ids, cos_sin_cache, kv_cache)
        return elementwise.add(hidden, delta.reshape(hidden.shape))


class TPTransformer:
    """Decoder-only LLM under tensor parallelism; ``model_type`` is set by the subclasses."""

    model_type = "dense"

    def __init__(self, model_config: ModelConfig, group=None, env=None) -> None:
        self.env = env
        self.dtype = model_config.dtype
        self.config = load_arch_config(model_config)
        c = self.config
        self.model_name = model_config.model_name
        self.max_length = model_config.max_length
        self.hidden_size = c.hidden_size
        self.num_heads = c.num_attention_heads
        self.head_dim = c.head_dim
        self.num_key_value_heads = c.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.max_position_embeddings = c.max_position_embeddings
        self.rope_theta = c.rope_theta
        self.rank, self.world_size, self.group = model_config.rank, model_config.world_size, group
        self.device = env.device if env is not None else (
            get_env().device if is_initialized() else
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self.layers: List[DecoderLayer] = []
        self.init_parameters()
        self.set_fwd()
        self.use_ar = False

    # ---- parameters ----------------------------------------------------------------------------
    def init_parameters(self):
        src = init_model_cpu(self.model_name, dtype=self.dtype, arch=self.config, device=self.device)
        dev = self.device
        self.embed_tokens = src.model.embed_tokens.weight.detach().to(dev)
        self.lm_head = src.lm_head.weight.detach().to(dev)
        self.norm_weight = src.model.norm.weight.detach().to(dev)
        self.norm_variance_epsilon = _eps(src.model.norm)
        self.cos_sin_cache = _set_cos_sin_cache(src.model.rotary_emb.inv_freq.to(dev), max_length=self.max_length)
        moe = self.model_type == "moe"
        for idx in range(self.config.num_hidden_layers):
            layer = DecoderLayer(idx, self.group, moe=moe, env=self.env)
            layer.init_parameters(hf_layer=src.model.layers[idx], rank=self.rank, world_size=self.world_size)
            self.layers.append(layer)
            src.model.layers[idx] = None
        self.num_layers = len(self.layers)

    def set_fwd(self, mode: str = "torch"):
        self.mode = mode
        for layer in self.layers:
            layer.set_fwd(mode)

    # ---- contexts (one per op kind, shared by every layer) -----------------------------------------
    def _share(self, names_attn, names_mlp):
        first = self.layers[0]
        for layer in self.layers[1:]:
            for n in names_attn:
                setattr(layer.attn, n, getattr(first.attn, n))
            for n in names_mlp:
                setattr(layer.mlp, n, getattr(first.mlp, n))

    def init_triton_dist_ctx(self, max_M: int = 4096):
        """``max_M``: rows entering AG+GEMM over ALL ranks (tokens of the whole batch)."""
        first = self.layers[0]
        first.attn._init_ctx(max_M=max_M)
        if self.model_type == "moe":
            first.mlp._init_ctx(M=max_M)
            self._share(("ag_ctx", "rs_ctx"), ("ag_ctx", "rs_ctx", "meta_ctx", "meta_buf", "meta_bytes", "meta_state",
                                               "max_M", "dec_per", "dec_x_ctx", "dec_x_buf", "dec_ar_ctx"))
        else:
            first.mlp._init_ctx(max_M=max_M)
            self._share(("ag_ctx", "rs_ctx"), ("ag_ctx", "rs_ctx"))
        self.use_ar = False

    def init_triton_dist_AR_ctx(self, max_M: int = 128, ar_method: AllReduceMethod = AllReduceMethod.Unknown):
        first = self.layers[0]
        first.attn._init_AR_ctx(max_M=max_M, method=ar_method, dtype=self.dtype)
        first.mlp._init_AR_ctx(max_M=max_M, method=ar_method, dtype=self.dtype)
        self._share(("ar_ctx", "ar_method"), ("ar_ctx", "ar_method"))
        self.use_ar = True

    def init_triton_dist_gemm_ar_ctx(self, max_M: int = 4096):
        first = self.layers[0]
        first.attn._init_gemm_ar_ctx(max_M=max_M, dtype=self.dtype)
        first.mlp._init_gemm_ar_ctx(max_M=max_M, dtype=self.dtype)
        self._share(("gemm_ar_ctx",), ("gemm_ar_ctx",))
        self.use_ar = True

    def finalize(self):
        for layer in self.layers:
            layer.attn.finalize()
            layer.mlp.finalize()

    # ---- forward -------------------------------------------------------------------------------
    @torch.inference_mode()
    def inference(self, input_ids: torch.LongTensor, position_ids: torch.LongTensor, kv_cache: KV_Cache,
                  wo_lm_head: bool = False):
        """input_ids [bsz, seq] (this rank's batch shard in ``triton_dist`` mode, the whole batch
        otherwise); position_ids [bsz_total, seq].  Returns fp32 logits of the last position."""
        seq_len = input_ids.shape[1]
        hidden = F.embedding(input_ids, self.embed_tokens)
        delta = None
        for layer in self.layers:
            hidden, delta = layer.fwd_fused(hidden, delta, position_ids, self.cos_sin_cache, kv_cache)
        if delta is None:
            hidden = elementwise.rms_norm(hidden, self.norm_weight, self.norm_variance_epsilon)
        else:
            hidden, _ = elementwise.rms_norm(delta.reshape(hidden.shape), self.norm_weight, self.norm_variance_epsilon,
                                             residual=hidden)
        if seq_len > 1:
            hidden = hidden[:, -1:]
        if wo_lm_head:
            return hidden
        return _linear(hidden.contiguous(), self.lm_head).float()
