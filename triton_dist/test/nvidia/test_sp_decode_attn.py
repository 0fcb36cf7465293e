"""Sequence-parallel GQA decode: the KV cache of every sequence is sharded over the ranks; each rank
decodes over its shard and the partials are merged through the low-latency all-gather.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_sp_decode_attn.py --kv-len 65536
    python triton_dist/test/nvidia/test_sp_decode_attn.py --virtual-ranks 4 --check

(CLI of reference python/triton_dist/test/nvidia/test_sp_decode_attn.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import math  # noqa: E402

from triton_dist.layers.nvidia.sp_flash_decode_layer import SpGQAFlashDecodeAttention  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--kv-len", type=int, default=4096, help="global KV length per sequence")
    ap.add_argument("--q-heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=8)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--page-size", type=int, default=16)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, dt = R.world_size, R.device, torch.bfloat16
    B, L, Hq, Hkv, D, P = args.batch, args.kv_len, args.q_heads, args.kv_heads, args.head_dim, args.page_size
    scale = 1.0 / math.sqrt(D)
    layers = R.collective(lambda r, env: SpGQAFlashDecodeAttention(r, 0, w, 1, Hq, Hkv, D, D, page_size=P, scale=scale,
                                                                   max_allowed_batch=B, env=env))
    g = torch.Generator(device="cpu").manual_seed(args.seed)
    q = torch.randn(B, Hq, D, generator=g).to(dev).to(dt)                      # identical on every rank
    kf = (torch.randn(B, L, Hkv, D, generator=g) * 0.5).to(dev).to(dt)
    vf = (torch.randn(B, L, Hkv, D, generator=g) * 0.5).to(dev).to(dt)
    per = (L + w - 1) // w
    lens = torch.tensor([[max(0, min(per, L - r * per)) for _ in range(B)] for r in range(w)], dtype=torch.int32,
                        device=dev)
    pages = (per + P - 1) // P
    caches = {}
    for r in R.local_ranks:
        k = torch.zeros(B * pages, P, Hkv, D, device=dev, dtype=dt)
        v = torch.zeros_like(k)
        n = int(lens[r, 0])
        k.view(B, pages * P, Hkv, D)[:, :n] = kf[:, r * per:r * per + n]
        v.view(B, pages * P, Hkv, D)[:, :n] = vf[:, r * per:r * per + n]
        caches[r] = (k, v, torch.arange(B * pages, device=dev, dtype=torch.int32).view(B, pages))
    R.barrier()
    fwd = lambda r, env: layers[r](q, caches[r][0], caches[r][1], lens, caches[r][2])   # noqa: E731
    if args.check:
        for it in range(3):
            outs = R.run(fwd)
            s = torch.einsum("bhgd,blhd->bhgl", q.float().view(B, Hkv, Hq // Hkv, D), kf.float()) * scale
            ref = torch.einsum("bhgl,blhd->bhgd", torch.softmax(s, -1), vf.float()).reshape(B, Hq, D)
            for r in R.local_ranks:
                err = (outs[r].float() - ref).abs().max().item()
                assert err < 2e-2, f"rank {r} iteration {it}: {err}"
        R.print(f"✅ sp flash decode batch={B} kv_len={L} world={w} check passed")
    else:
        ms = R.time(fwd, args.warmup, args.iters)
        R.print_all(f"sp flash decode #ranks {w}: {ms * 1e3:.1f} us (kv_len {L}, batch {B})")
    R.close()


if __name__ == "__main__":
    main()
