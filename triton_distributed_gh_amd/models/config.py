"""Model configuration.

``ModelConfig`` is the user-facing handle (reference: python/triton_dist/models/config.py:29-37).
``ArchConfig`` carries the architecture hyper-parameters the reference reads from
``transformers.AutoConfig`` (models/dense.py:126-137).  This build targets machines without network
access, so the architectures of every entry of the reference's model table
(models/__init__.py:34-43) are compiled in; a locally cached HuggingFace config is preferred when
one exists.
"""
from __future__ import annotations

import dataclasses
import os
from typing import Optional

import torch


@dataclasses.dataclass
class ModelConfig:
    model_name: str = "Qwen/Qwen3-32B"
    max_length: int = 4096
    dtype: torch.dtype = torch.bfloat16
    local_only: bool = False
    rank: int = 0
    world_size: int = 8
    # extensions (not in the reference): truncate the stack for smoke tests / single-GPU benches
    num_layers: Optional[int] = None
    arch: Optional["ArchConfig"] = None


@dataclasses.dataclass
class ArchConfig:
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int = 128
    vocab_size: int = 151936
    rope_theta: float = 1e6
    rms_norm_eps: float = 1e-6
    max_position_embeddings: int = 40960
    qk_norm: bool = True
    tie_word_embeddings: bool = False
    # MoE
    num_experts: int = 0
    num_experts_per_tok: int = 0
    moe_intermediate_size: int = 0
    norm_topk_prob: bool = True

    @property
    def is_moe(self) -> bool:
        return self.num_experts > 0


KNOWN_ARCHS = {
    "Qwen/Qwen3-0.6B": ArchConfig(1024, 3072, 28, 16, 8, tie_word_embeddings=True),
    "Qwen/Qwen3-1.7B": ArchConfig(2048, 6144, 28, 16, 8, tie_word_embeddings=True),
    "Qwen/Qwen3-4B": ArchConfig(2560, 9728, 36, 32, 8, tie_word_embeddings=True),
    "Qwen/Qwen3-8B": ArchConfig(4096, 12288, 36, 32, 8),
    "Qwen/Qwen3-14B": ArchConfig(5120, 17408, 40, 40, 8),
    "Qwen/Qwen3-32B": ArchConfig(5120, 25600, 64, 64, 8),
    "Qwen/Qwen3-30B-A3B": ArchConfig(2048, 6144, 48, 32, 4, num_experts=128, num_experts_per_tok=8,
                                     moe_intermediate_size=768),
    "Qwen/Qwen3-235B-A22B": ArchConfig(4096, 12288, 94, 64, 4, num_experts=128, num_experts_per_tok=8,
                                       moe_intermediate_size=1536),
    "meta-llama/Meta-Llama-3-70B": ArchConfig(8192, 28672, 80, 64, 8, vocab_size=128256, rope_theta=5e5,
                                              rms_norm_eps=1e-5, max_position_embeddings=8192, qk_norm=False),
    "ByteDance-Seed/Seed-OSS-36B-Instruct": ArchConfig(5120, 27648, 64, 80, 8, vocab_size=155136, rope_theta=1e7,
                                                       max_position_embeddings=524288, qk_norm=False),
    # tiny architectures for tests and smoke runs (not in the reference's table)
    "tdist/tiny-dense": ArchConfig(512, 1024, 2, 8, 8, head_dim=64, vocab_size=1024, max_position_embeddings=2048),
    "tdist/tiny-dense-d128": ArchConfig(1024, 2048, 2, 8, 4, head_dim=128, vocab_size=1024, max_position_embeddings=2048),
    "tdist/tiny-moe": ArchConfig(512, 1024, 2, 8, 8, head_dim=64, vocab_size=1024, max_position_embeddings=2048,
                                 num_experts=8, num_experts_per_tok=2, moe_intermediate_size=1024),
}


def _from_hf(model_name: str, local_only: bool) -> Optional[ArchConfig]:
    if os.environ.get("TDIST_NO_HF_CONFIG", "0") == "1":
        return None
    try:
        from transformers import AutoConfig

        c = AutoConfig.from_pretrained(model_name, local_files_only=True)
    except Exception:
        return None
    head_dim = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
    return ArchConfig(
        hidden_size=c.hidden_size, intermediate_size=c.intermediate_size, num_hidden_layers=c.num_hidden_layers,
        num_attention_heads=c.num_attention_heads, num_key_value_heads=c.num_key_value_heads, head_dim=head_dim,
        vocab_size=c.vocab_size, rope_theta=float(getattr(c, "rope_theta", 1e4)),
        rms_norm_eps=float(getattr(c, "rms_norm_eps", 1e-6)),
        max_position_embeddings=int(getattr(c, "max_position_embeddings", 4096)),
        qk_norm="qwen3" in getattr(c, "model_type", ""),
        tie_word_embeddings=bool(getattr(c, "tie_word_embeddings", False)),
        num_experts=int(getattr(c, "num_experts", 0) or 0),
        num_experts_per_tok=int(getattr(c, "num_experts_per_tok", 0) or 0),
        moe_intermediate_size=int(getattr(c, "moe_intermediate_size", 0) or 0),
        norm_topk_prob=bool(getattr(c, "norm_topk_prob", True)))


def load_arch_config(model_config: ModelConfig) -> ArchConfig:
    arch = model_config.arch or _from_hf(model_config.model_name, model_config.local_only) \
        or KNOWN_ARCHS.get(model_config.model_name)
    if arch is None:
        raise KeyError(f"unknown model {model_config.model_name!r}; known: {sorted(KNOWN_ARCHS)} "
                       "(or pass ModelConfig(arch=ArchConfig(...)))")
    if model_config.num_layers is not None:
        arch = dataclasses.replace(arch, num_hidden_layers=int(model_config.num_layers))
    return arch
