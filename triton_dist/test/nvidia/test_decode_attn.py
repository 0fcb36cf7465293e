"""Single-GPU paged GQA flash decode: split-KV + merge, fused one-launch, and the LSE-emitting
intra-rank form - correctness vs fp32 PyTorch and latency / KV bandwidth.

    python triton_dist/test/nvidia/test_decode_attn.py --batch 8 --kv-len 8192 --q-heads 32 --kv-heads 8
    python triton_dist/test/nvidia/test_decode_attn.py --check

(CLI of reference python/triton_dist/test/nvidia/test_decode_attn.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import math  # noqa: E402
import time  # noqa: E402

from triton_dist.kernels.nvidia import flash_decode as fd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--kv-len", type=int, default=4096)
    ap.add_argument("--q-heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=8)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--page-size", type=int, default=16)
    ap.add_argument("--soft-cap", type=float, default=0.0)
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float16"])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    dt = getattr(torch, args.dtype)
    B, L, Hq, Hkv, D, P = args.batch, args.kv_len, args.q_heads, args.kv_heads, args.head_dim, args.page_size
    pages = (L + P - 1) // P + 1
    k = (torch.randn(B * pages, P, Hkv, D, device=dev) * 0.5).to(dt)
    v = (torch.randn(B * pages, P, Hkv, D, device=dev) * 0.5).to(dt)
    table = torch.randperm(B * pages, device=dev).int().view(B, pages).contiguous()
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    lens[0] = max(1, L // 3)
    q = torch.randn(B, Hq, D, device=dev).to(dt)
    scale = 1.0 / math.sqrt(D)
    variants = {"split+combine": fd.gqa_fwd_batch_decode, "persistent": fd.gqa_fwd_batch_decode_persistent,
                "intra_rank": fd.gqa_fwd_batch_decode_intra_rank}
    if args.check:
        ref, ref_lse = fd.decode_reference(q, k, v, lens, table, scale, args.soft_cap, return_lse=True)
        tol = 2e-2 if dt == torch.bfloat16 else 4e-3
        for name, fn in variants.items():
            o = fn(q, k, v, None, None, lens, table, scale, args.soft_cap)
            err = (o[..., :D].float() - ref).abs().max().item()
            assert err < tol, f"{name}: {err}"
            if name == "intra_rank":
                assert (o[..., D] - ref_lse).abs().max().item() < 2e-2
        print(f"✅ flash decode batch={B} kv_len={L} heads={Hq}/{Hkv} D={D} page={P} check passed", flush=True)
        return
    for name, fn in variants.items():
        for _ in range(5):
            fn(q, k, v, None, None, lens, table, scale, args.soft_cap)
        if dev == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn(q, k, v, None, None, lens, table, scale, args.soft_cap)
        if dev == "cuda":
            torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / args.iters * 1e6
        kv_bytes = 2.0 * float(lens.sum()) * Hkv * D * k.element_size()
        print(f"{name:>14}: {us:8.1f} us  ({kv_bytes / us / 1e6:.2f} TB/s of KV)", flush=True)


if __name__ == "__main__":
    main()
