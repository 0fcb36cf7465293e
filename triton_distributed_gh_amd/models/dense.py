"""Dense decoder LLMs (Qwen3 dense, Llama-3, Seed-OSS) under tensor parallelism
(reference: python/triton_dist/models/dense.py:52-241)."""
from __future__ import annotations

from .transformer import DecoderLayer, TPTransformer


class DenseLLMLayer(DecoderLayer):

    def __init__(self, layer_idx, group, env=None) -> None:
        super().__init__(layer_idx, group, moe=False, env=env)


class DenseLLM(TPTransformer):
    model_type = "dense"
