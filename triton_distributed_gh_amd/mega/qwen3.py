"""Qwen3-style dense decoder as ONE launch per decode step: a replayed HIP graph of fused kernels (``backend="auto"``
/ ``"graph"``, the default and the faster form on MI355X) or the persistent interpreter kernel (``"interpreter"``).

Reference: python/triton_dist/mega_triton_kernel/models/qwen3.py (``Qwen3Model`` :191-201
``mega_forwrad``), models/layers/tp_attn.py, tp_mlp.py and models/paged_kv_cache.py.  The graph is
built once from the tensor-parallel shards of a ``DenseLLM``; every step writes the token embeddings
into the input buffer, launches the interpreter kernel and bumps the KV lengths.

Per layer: [add +] rms_norm -> qkv_proj -> qk_norm + rope + kv-cache update -> split-KV attention ->
combine -> o_proj [-> all-reduce] -> add + rms_norm -> fc1 (gate/up GEMM with fused SiLU*up) -> fc2
[-> all-reduce]; residual adds are folded into the norm tasks."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from ..layers.tp_mlp import shard_local  # noqa: F401  (reference: models/qwen3.py:35)
from ..models.kv_cache import PagedKVCache
from .layers import TPAttnBuilder, TPMLPBuilder
from .model_builder import ModelBuilder


class Qwen3LayerBuilder:
    """One decoder layer recorded from a HuggingFace-layout module (``self_attn`` / ``mlp`` / ``input_layernorm`` /
    ``post_attention_layernorm``): reference models/qwen3.py:50-110.  ``build_fwd`` takes the residual stream
    [tokens, H] and returns the tensor holding the updated stream (the same buffer, updated in place by the fused
    add + norm tasks)."""

    def __init__(self, builder: ModelBuilder, layer_idx: int, head_dim: int, rank: int = 0, world_size: int = 1):
        self._builder, self.layer_idx, self.head_dim, self.rank, self.world_size = builder, layer_idx, head_dim, rank, world_size
        self.attn: Optional[TPAttnBuilder] = None
        self.mlp: Optional[TPMLPBuilder] = None
        self.input_norm_w = self.post_norm_w = None
        self.input_norm_eps = self.post_norm_eps = 1e-6

    def init_parameters(self, hf_layer):
        dev = self._builder.device
        self.mlp = TPMLPBuilder(self._builder, self.rank, self.world_size)._init_parameters(hf_layer.mlp)
        self.attn = TPAttnBuilder(self._builder, self.layer_idx, self.head_dim, self.rank,
                                  self.world_size)._init_parameters(hf_layer.self_attn)
        n1, n2 = hf_layer.input_layernorm, hf_layer.post_attention_layernorm
        self.input_norm_w, self.post_norm_w = n1.weight.detach().to(dev), n2.weight.detach().to(dev)
        self.input_norm_eps = float(getattr(n1, "variance_epsilon", getattr(n1, "eps", 1e-6)))
        self.post_norm_eps = float(getattr(n2, "variance_epsilon", getattr(n2, "eps", 1e-6)))
        return self

    def build_fwd(self, hidden_states: torch.Tensor, cos_cache: torch.Tensor, sin_cache: torch.Tensor, kv_cache):
        b, li = self._builder, self.layer_idx
        shape = hidden_states.shape
        h = hidden_states.view(-1, shape[-1])
        xn = torch.zeros_like(h)
        b.make_rms_norm(h, self.input_norm_w, xn, self.input_norm_eps, name=f"l{li}.norm1")
        a = self.attn.build_fwd(xn, cos_cache, sin_cache, kv_cache, name=f"l{li}")
        b.make_rms_norm(a, self.post_norm_w, xn, self.post_norm_eps, residual=h, residual_out=h, name=f"l{li}.norm2")
        m = self.mlp.build_fwd(xn, name=f"l{li}")
        b.make_add(h, m, h, name=f"l{li}.residual")
        return hidden_states


class Qwen3Model:

    def __init__(self, model, batch_size: int, max_length: Optional[int] = None, page_size: int = 64,
                 num_queues: int = 256, scheduler: str = "round_robin", env=None, prefetch_weights: bool = False,
                 backend: str = "auto"):
        assert batch_size <= 16, "decode batch of at most 16 tokens per step"
        self.model, self.batch_size = model, batch_size
        self.env = env or model.env
        w, dev, dt = model.world_size, model.device, model.dtype
        assert dt == torch.bfloat16, "mega-kernel tasks are bf16"
        self.world_size = w
        H, D = model.hidden_size, model.head_dim
        Hq, Hkv = model.num_heads // w, model.num_key_value_heads // w
        self.max_length = max_length or model.max_length
        self.kv_cache = PagedKVCache(model.num_layers, batch_size, self.max_length, model.num_key_value_heads, D,
                                     page_size=page_size, dtype=dt, world_size=w, device=dev)
        self.kv_lens = self.kv_cache.kv_offset
        b = self.builder = ModelBuilder(model.rank, w, w, num_queues=num_queues, scheduler=scheduler, env=self.env)
        B = batch_size
        new = lambda *s: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
        symm = (lambda *s: b.create_symm_tensor(s, dt)) if w > 1 else new
        self.hidden = new(B, H)                  # residual stream (input: token embeddings)
        xn, qkv = new(B, H), new(B, (Hq + 2 * Hkv) * D)
        q, attn = new(B, Hq, D), new(B, Hq, D)
        o_buf, mlp_buf = symm(B, H), symm(B, H)
        o_red = new(B, H) if w > 1 else o_buf
        mlp_red = new(B, H) if w > 1 else mlp_buf
        inter = model.layers[0].mlp.gate_up_proj.shape[0] // 2
        act = new(B, inter)
        cos, sin = model.cos_sin_cache
        delta = None
        self.attn_builders, self.mlp_builders = [], []
        for li, layer in enumerate(model.layers):
            a, m = layer.attn, layer.mlp
            ab = TPAttnBuilder(b, li, D, model.rank, w).from_layer(a, Hq, Hkv)
            mb = TPMLPBuilder(b, model.rank, w).from_layer(m)
            self.attn_builders.append(ab)
            self.mlp_builders.append(mb)
            if prefetch_weights:
                b.make_prefetch(m.gate_up_proj, name=f"l{li}.prefetch_fc1")
            b.make_rms_norm(self.hidden if delta is None else delta, layer.input_norm_w, xn, layer.input_norm_eps,
                            residual=None if delta is None else self.hidden,
                            residual_out=None if delta is None else self.hidden, name=f"l{li}.norm1")
            # (the activation buffers are shared by all layers: every task that reuses one depends, through the
            # chain of ops in between, on the last reader of its previous contents)
            o_red_l = ab.build_fwd(xn, cos, sin, self.kv_cache, qkv=qkv, q=q, attn=attn, o_out=o_buf,
                                   ar_out=o_red if w > 1 else None, name=f"l{li}")
            b.make_rms_norm(o_red_l, layer.post_norm_w, xn, layer.post_norm_eps, residual=self.hidden,
                            residual_out=self.hidden, name=f"l{li}.norm2")
            delta = mb.build_fwd(xn, act_out=act, fc2_out=mlp_buf, ar_out=mlp_red if w > 1 else None, name=f"l{li}")
        self.final = new(B, H)
        b.make_rms_norm(delta, model.norm_weight, self.final, model.norm_variance_epsilon, residual=self.hidden,
                        residual_out=self.hidden, name="final_norm")
        V = model.lm_head.shape[0]
        self.logits = new(B, V)
        b.make_linear(self.final, model.lm_head, self.logits, name="lm_head")
        self.compiled = b.compile(backend=backend, outputs=[self.logits, self.hidden])
        self.backend = self.compiled["backend"]

    @property
    def num_tasks(self) -> int:
        return self.compiled["num_tasks"]

    def set_kv_len(self, n: int):
        self.kv_lens.fill_(n)

    @torch.inference_mode()
    def mega_forward(self, input_ids: torch.Tensor) -> torch.Tensor:
        """input_ids [B, 1] -> fp32 logits [B, 1, V]; appends the tokens to the KV cache."""
        self.hidden.copy_(F.embedding(input_ids.view(-1), self.model.embed_tokens))
        self.builder.run()
        self.kv_lens += 1
        return self.logits.float().unsqueeze(1)

    mega_forwrad = mega_forward     # the reference's spelling

    def check_error(self):
        self.builder.check_error()
