"""Qwen3 Mixture-of-Experts models under tensor parallelism (every expert's FFN sharded over the TP
ranks; reference: python/triton_dist/models/qwen_moe.py:50-206, which has no AMD path)."""
from __future__ import annotations

from .transformer import DecoderLayer, TPTransformer


class Qwen3MoELayer(DecoderLayer):

    def __init__(self, layer_idx, group, env=None) -> None:
        super().__init__(layer_idx, group, moe=True, env=env)


class Qwen3MoE(TPTransformer):
    model_type = "moe"
