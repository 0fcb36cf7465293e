"""Parallelism strategies, grouped: the building block(s) that implement each one.

``tp``  tensor parallel (column-parallel AG+GEMM, row-parallel GEMM+RS / GEMM+AR, sequence-sharded activations)
``ep``  expert parallel (low-latency all-to-all for decode, dispatch / combine for large batches)
``sp``  sequence parallel (KV sharded flash-decode with LSE merge; all-gather-KV prefill attention)
``pp``  pipeline parallel primitive (symmetric send / recv buffers + signals)
(reference: the table of strategies realised by python/triton_dist/layers/{amd,nvidia}/*.py)
"""
from types import SimpleNamespace

from ..layers import (AllGatherLayer, CommOp, EPAll2AllLayer, GemmARLayer, SpGQAFlashDecodeAttention, TP_Attn, TP_MLP,
                      TP_MoE)
from ..ops.all_to_all import create_all_to_all_context, fast_all_to_all
from ..ops.sp_ag_attention import create_sp_ag_attention_context_intra_node, fused_sp_ag_attn_intra_node

tp = SimpleNamespace(TP_MLP=TP_MLP, TP_Attn=TP_Attn, TP_MoE=TP_MoE, GemmARLayer=GemmARLayer)
ep = SimpleNamespace(EPAll2AllLayer=EPAll2AllLayer, create_all_to_all_context=create_all_to_all_context,
                     fast_all_to_all=fast_all_to_all)
sp = SimpleNamespace(SpGQAFlashDecodeAttention=SpGQAFlashDecodeAttention, AllGatherLayer=AllGatherLayer,
                     create_sp_ag_attention_context_intra_node=create_sp_ag_attention_context_intra_node,
                     fused_sp_ag_attn_intra_node=fused_sp_ag_attn_intra_node)
pp = SimpleNamespace(CommOp=CommOp)
__all__ = ["tp", "ep", "sp", "pp"]
