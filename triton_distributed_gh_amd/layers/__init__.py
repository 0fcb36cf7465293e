"""Model building blocks over the fused ops (reference: python/triton_dist/layers/{amd,nvidia}/)."""
from ..ops.ep_a2a import EPAll2AllLayer  # noqa: F401
from .gemm_allreduce_layer import GemmARLayer  # noqa: F401
from .low_latency_allgather_layer import AllGatherLayer  # noqa: F401
from .p2p import CommOp  # noqa: F401
from .sp_flash_decode_layer import SpGQAFlashDecodeAttention  # noqa: F401
from .tp_attn import TP_Attn, _set_cos_sin_cache, apply_rotary_pos_emb, layer_norm  # noqa: F401
from .tp_mlp import TP_MLP, shard_local  # noqa: F401
from .tp_moe import TP_MoE  # noqa: F401
