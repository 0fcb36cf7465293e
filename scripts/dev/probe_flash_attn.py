"""Prefill attention (varlen causal GQA) throughput of the in-tree kernel vs torch SDPA."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from triton_distributed_gh_amd.ops.flash_attn import flash_attn_varlen_fwd

def bench(fn, iters=20, warmup=5):
    for _ in range(warmup): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for S, Hq, Hkv, D in [(2048, 32, 8, 128), (4096, 32, 8, 128), (8192, 32, 8, 128), (4096, 64, 8, 128), (4096, 32, 8, 64),
                      (4096, 8, 1, 128), (2048, 8, 2, 128), (8192, 8, 1, 128), (1024, 32, 8, 128), (4096, 4, 1, 128), (4096, 12, 4, 128), (4096, 20, 4, 128)]:
    q = (torch.randn(S, Hq, D, device="cuda") * 0.5).bfloat16()
    k = (torch.randn(S, Hkv, D, device="cuda") * 0.5).bfloat16()
    v = (torch.randn(S, Hkv, D, device="cuda") * 0.5).bfloat16()
    cu = torch.tensor([0, S], dtype=torch.int32, device="cuda")
    t = bench(lambda: flash_attn_varlen_fwd(q, k, v, cu, cu, S, causal=True))
    flops = 4.0 * S * S * Hq * D / 2
    qs, ks, vs = q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None]
    try:
        tt = bench(lambda: torch.nn.functional.scaled_dot_product_attention(qs, ks, vs, is_causal=True, enable_gqa=True))
    except Exception as e:  # noqa: BLE001
        tt = float("nan")
    print(f"S={S} Hq={Hq} Hkv={Hkv} D={D}: native {t:8.1f} us ({flops / t / 1e6:6.1f} TF)   torch sdpa {tt:8.1f} us ({flops / tt / 1e6:6.1f} TF)", flush=True)
