"""Sequence-parallel prefill attention with all-gathered KV (zig-zag or contiguous shards).

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_sp_ag_attention_intra_node.py --seqlens 8192 16384
    python triton_dist/test/nvidia/test_sp_ag_attention_intra_node.py --virtual-ranks 4 --check

(CLI of reference python/triton_dist/test/nvidia/test_sp_ag_attention_intra_node.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.nvidia.sp_ag_attention_intra_node import (attention_reference,  # noqa: E402
                                                                   create_sp_ag_attention_context_intra_node,
                                                                   fused_sp_ag_attn_intra_node, shard_positions)
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqlens", type=int, nargs="+", default=[1024, 2048], help="full length of every sequence")
    ap.add_argument("--q-heads", type=int, default=8)
    ap.add_argument("--kv-heads", type=int, default=2)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--no-zig-zag", action="store_true")
    ap.add_argument("--no-causal", action="store_true")
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, dt = R.world_size, R.device, torch.bfloat16
    Hq, Hkv, D = args.q_heads, args.kv_heads, args.head_dim
    zig, causal = not args.no_zig_zag, not args.no_causal
    full = [((L + 128 * w - 1) // (128 * w)) * 128 * w for L in args.seqlens]      # 64-row half blocks per rank
    shard = [L // w for L in full]
    nb = len(full)
    cu_k = torch.tensor([0] + list(torch.tensor(full).cumsum(0)), dtype=torch.int32, device=dev)
    cu_q = torch.tensor([0] + list(torch.tensor(shard).cumsum(0)), dtype=torch.int32, device=dev)
    ctxs = R.collective(lambda r, env: create_sp_ag_attention_context_intra_node(
        nb, Hq, Hkv, max(full), max(shard), D, dt, dt, r, w, env=env))
    g = torch.Generator(device="cpu").manual_seed(args.seed)
    qf = torch.randn(sum(full), Hq, D, generator=g).to(dev).to(dt)
    kf = torch.randn(sum(full), Hkv, D, generator=g).to(dev).to(dt)
    vf = torch.randn(sum(full), Hkv, D, generator=g).to(dev).to(dt)

    def rows(r):
        return torch.cat([shard_positions([0, shard[b]], r, w, zig) + int(cu_k[b]) for b in range(nb)]).to(dev)

    qs = {r: qf[rows(r)].contiguous() for r in R.local_ranks}
    ks = {r: kf[rows(r)].contiguous() for r in R.local_ranks}
    vs = {r: vf[rows(r)].contiguous() for r in R.local_ranks}
    outs = {r: torch.empty_like(qs[r]) for r in R.local_ranks}
    R.barrier()
    fwd = lambda r, env: fused_sp_ag_attn_intra_node(ctxs[r], qs[r], ks[r], vs[r], outs[r], cu_q, cu_k,   # noqa: E731
                                                     max(shard), max(full), r, w, causal, zig)
    if args.check:
        ref = attention_reference(qf, kf, vf, cu_k.tolist(), cu_k.tolist(), causal)
        for it in range(3):
            R.run(fwd)
            for r in R.local_ranks:
                ctxs[r].check_error()
                err = (outs[r].float() - ref[rows(r)]).abs().max().item()
                assert err < 3e-2, f"rank {r} iteration {it}: {err}"
        R.print(f"✅ sp_ag_attention seqlens={full} heads={Hq}/{Hkv} zigzag={zig} causal={causal} world={w} check passed")
    else:
        ms = R.time(fwd, args.warmup, args.iters)
        fl = sum(4.0 * L * L * Hq * D * (0.5 if causal else 1.0) for L in full) / w
        R.print_all(f"sp_ag_attention #ranks {w}: {ms:.4f} ms ({fl / ms / 1e9:.1f} TFLOP/s per rank)")
    R.close()


if __name__ == "__main__":
    main()
