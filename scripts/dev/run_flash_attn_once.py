"""Causal GQA prefill attention, S=4096 Hq=32 Hkv=8 D=128 bf16: a few calls of the native kernel
(and of torch SDPA with `sdpa` as the first argument) for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from triton_distributed_gh_amd.ops.flash_attn import flash_attn_varlen_fwd

S, Hq, Hkv, D = 4096, 32, 8, 128
q = (torch.randn(S, Hq, D, device="cuda") * 0.5).bfloat16()
k = (torch.randn(S, Hkv, D, device="cuda") * 0.5).bfloat16()
v = (torch.randn(S, Hkv, D, device="cuda") * 0.5).bfloat16()
cu = torch.tensor([0, S], dtype=torch.int32, device="cuda")
qs, ks, vs = q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None]
for _ in range(8):
    if len(sys.argv) > 1 and sys.argv[1] == "sdpa":
        torch.nn.functional.scaled_dot_product_attention(qs, ks, vs, is_causal=True, enable_gqa=True)
    else:
        flash_attn_varlen_fwd(q, k, v, cu, cu, S, causal=True)
torch.cuda.synchronize()
