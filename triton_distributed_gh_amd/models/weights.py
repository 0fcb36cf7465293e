"""Weight sources shaped like a HuggingFace causal-LM module tree.

The reference materialises ``transformers.AutoModelForCausalLM`` (pre-trained, or random when the
``RANDOM_PARAMS`` env var is set - reference: python/triton_dist/models/utils.py:107-128) and then
shards it layer by layer.  Holding a whole 32B model on the host only to shard it is wasteful, so the
random source here generates ONE decoder layer at a time, directly on the target device, from
per-parameter seeds: every rank produces bit-identical full tensors and keeps only its shard.

The tree exposes the attribute names the layers read: ``model.embed_tokens.weight``,
``model.layers[i].self_attn.{q,k,v,o}_proj.weight`` (+ ``q_norm`` / ``k_norm``),
``model.layers[i].mlp.{gate,up,down}_proj.weight`` or the MoE form ``mlp.gate.weight`` /
``mlp.experts[e]``, ``input_layernorm`` / ``post_attention_layernorm`` (``weight``,
``variance_epsilon``), ``model.norm``, ``model.rotary_emb.inv_freq`` and ``lm_head.weight``.
"""
from __future__ import annotations

import os
import zlib
from types import SimpleNamespace
from typing import Iterator

import torch

from .config import ArchConfig


def _seed(*key) -> int:
    return zlib.crc32("/".join(str(k) for k in key).encode()) & 0x7FFFFFFF


class RandomWeights:
    """Lazy, deterministic random-init model ("RANDOM_PARAMS" behaviour of the reference)."""

    def __init__(self, arch: ArchConfig, dtype: torch.dtype, device, seed: int = 1234, std: float = 0.02):
        self.arch, self.dtype, self.device, self.seed, self.std = arch, dtype, torch.device(device), seed, std
        a = arch
        inv_freq = 1.0 / (a.rope_theta ** (torch.arange(0, a.head_dim, 2, dtype=torch.float32, device=self.device)
                                           / a.head_dim))
        embed = self._linear("embed", a.vocab_size, a.hidden_size)
        self.model = SimpleNamespace(
            embed_tokens=embed,
            norm=self._norm("final_norm", a.hidden_size),
            rotary_emb=SimpleNamespace(inv_freq=inv_freq),
            layers=_LazyLayers(self),
        )
        self.lm_head = embed if a.tie_word_embeddings else self._linear("lm_head", a.vocab_size, a.hidden_size)

    # ---- generators --------------------------------------------------------------------------------
    def _randn(self, name, *shape) -> torch.Tensor:
        g = torch.Generator(device=self.device)
        g.manual_seed(_seed(self.seed, name))
        t = torch.empty(*shape, dtype=torch.float32 if self.device.type == "cpu" else self.dtype, device=self.device)
        t.normal_(0.0, self.std, generator=g)
        return t.to(self.dtype)

    def _linear(self, name, out_f, in_f):
        return SimpleNamespace(weight=self._randn(name, out_f, in_f), bias=None)

    def _norm(self, name, n):
        w = 1.0 + self._randn(name, n).float()
        return SimpleNamespace(weight=w.to(self.dtype), variance_epsilon=self.arch.rms_norm_eps,
                               eps=self.arch.rms_norm_eps)

    def make_layer(self, idx: int):
        a = self.arch
        p = f"layer{idx}"
        q_out, kv_out = a.num_attention_heads * a.head_dim, a.num_key_value_heads * a.head_dim
        attn = SimpleNamespace(
            head_dim=a.head_dim,
            q_proj=self._linear(f"{p}.q", q_out, a.hidden_size),
            k_proj=self._linear(f"{p}.k", kv_out, a.hidden_size),
            v_proj=self._linear(f"{p}.v", kv_out, a.hidden_size),
            o_proj=self._linear(f"{p}.o", a.hidden_size, q_out),
            q_norm=self._norm(f"{p}.qn", a.head_dim) if a.qk_norm else None,
            k_norm=self._norm(f"{p}.kn", a.head_dim) if a.qk_norm else None,
        )
        if a.is_moe:
            mlp = SimpleNamespace(
                num_experts=a.num_experts, top_k=a.num_experts_per_tok, norm_topk_prob=a.norm_topk_prob,
                gate=self._linear(f"{p}.router", a.num_experts, a.hidden_size),
                experts=_LazyExperts(self, p),
            )
        else:
            mlp = self._ffn(p, a.intermediate_size)
        return SimpleNamespace(self_attn=attn, mlp=mlp, input_layernorm=self._norm(f"{p}.ln1", a.hidden_size),
                               post_attention_layernorm=self._norm(f"{p}.ln2", a.hidden_size))

    def _ffn(self, prefix, inter):
        h = self.arch.hidden_size
        return SimpleNamespace(gate_proj=self._linear(f"{prefix}.gate", inter, h),
                               up_proj=self._linear(f"{prefix}.up", inter, h),
                               down_proj=self._linear(f"{prefix}.down", h, inter),
                               act_fn=torch.nn.functional.silu)


class _LazyLayers:

    def __init__(self, src: RandomWeights):
        self.src = src

    def __len__(self):
        return self.src.arch.num_hidden_layers

    def __getitem__(self, idx):
        return self.src.make_layer(idx)

    def __setitem__(self, idx, value):  # the reference drops layers after sharding them
        pass

    def __iter__(self) -> Iterator:
        for i in range(len(self)):
            yield self.src.make_layer(i)


class _LazyExperts:

    def __init__(self, src: RandomWeights, prefix: str):
        self.src, self.prefix = src, prefix

    def __len__(self):
        return self.src.arch.num_experts

    def __getitem__(self, e):
        return self.src._ffn(f"{self.prefix}.expert{e}", self.src.arch.moe_intermediate_size)


def wants_random_params() -> bool:
    return os.environ.get("RANDOM_PARAMS", "0").lower() in ("1", "true", "yes")


def load_weight_source(model_name: str, arch: ArchConfig, dtype: torch.dtype, device, local_only: bool = True):
    """Pre-trained weights from the local HuggingFace cache when present, otherwise (or with
    ``RANDOM_PARAMS=1``) the lazy random source."""
    if not wants_random_params():
        try:
            from transformers import AutoModelForCausalLM

            return AutoModelForCausalLM.from_pretrained(model_name, torch_dtype=dtype, local_files_only=True)
        except Exception:
            pass
    return RandomWeights(arch, dtype, device)
