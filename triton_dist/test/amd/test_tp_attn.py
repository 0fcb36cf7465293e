"""TP attention layer (QKV AG+GEMM -> norm/rope -> flash attention -> O GEMM+RS): torch forward vs
the fused forwards, prefill then one decode step.

    bash scripts/launch_amd.sh python/triton_dist/test/amd/test_tp_attn.py --bsz 32 --seq_len 128 --model Qwen/Qwen3-32B
    python triton_dist/test/amd/test_tp_attn.py --model tdist/tiny-dense --bsz 4 --seq_len 16 --virtual-ranks 2 --check

(CLI of reference python/triton_dist/test/amd/test_tp_attn.py; layer 0 of the model, random weights.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
os.environ.setdefault("RANDOM_PARAMS", "1")

import torch  # noqa: E402

from triton_dist.layers.amd.tp_attn import TP_Attn, _set_cos_sin_cache  # noqa: E402
from triton_dist.models.kv_cache import KV_Cache  # noqa: E402
from triton_distributed_gh_amd.models.config import ModelConfig, load_arch_config  # noqa: E402
from triton_distributed_gh_amd.models.weights import RandomWeights  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402

MODES = {"ag_rs": "dist_triton_fwd", "allreduce": "dist_triton_AR_fwd", "gemm_ar": "dist_triton_gemm_ar_fwd"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bsz", type=int, default=32)
    ap.add_argument("--seq_len", type=int, default=128)
    ap.add_argument("--model", default="Qwen/Qwen3-32B")
    ap.add_argument("--mode", default="ag_rs", choices=list(MODES))
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev = R.world_size, R.device
    dt = torch.bfloat16 if dev.type == "cuda" else torch.float32
    B, S = args.bsz, args.seq_len
    arch = load_arch_config(ModelConfig(model_name=args.model))
    tree = RandomWeights(arch, dt, dev).make_layer(0).self_attn
    H, D, Hkv = arch.hidden_size, arch.head_dim, arch.num_key_value_heads
    L = S + 8
    inv = 1.0 / (arch.rope_theta ** (torch.arange(0, D, 2, device=dev).float() / D))
    cos_sin = _set_cos_sin_cache(inv, L)
    attns = {r: TP_Attn(rank=r, world_size=w, group=None if R.virtual else R.envs[0].tp_group,
                        env=R.envs[r] if R.virtual else R.envs[0]) for r in R.local_ranks}
    for a in attns.values():
        a._init_parameters(tree)

    def init(r, env):
        attns[r]._init_ctx(max_M=B * S)
        if args.mode == "allreduce":
            attns[r]._init_AR_ctx(max_M=B * S, dtype=dt)
        if args.mode == "gemm_ar":
            attns[r]._init_gemm_ar_ctx(max_M=B * S, dtype=dt)

    R.collective(init)
    torch.manual_seed(args.seed)
    x = (torch.rand(B, S, H, device=dev) * 2 - 1).to(dt)
    x1 = (torch.rand(B, 1, H, device=dev) * 2 - 1).to(dt)
    pos = torch.arange(S, device=dev)[None].repeat(B, 1)
    pos1 = torch.full((B, 1), S, device=dev, dtype=torch.long)
    sharded = args.mode == "ag_rs"
    per = B // w
    sh = (lambda t, r: t[r * per:(r + 1) * per]) if sharded else (lambda t, r: t)
    kvs = {r: KV_Cache(1, B, L, Hkv, D, dt, w, device=dev) for r in R.local_ranks}
    fn = MODES[args.mode]
    if args.check:
        gold = TP_Attn(rank=0, world_size=1, env=R.envs[R.local_ranks[0]] if R.virtual else R.envs[0])
        gold._init_parameters(tree)
        gkv = KV_Cache(1, B, L, Hkv, D, dt, 1, device=dev)
        ref = gold.torch_fwd(x, pos, cos_sin, gkv, 0)
        gkv.kv_offset.fill_(S)
        ref1 = gold.torch_fwd(x1, pos1, cos_sin, gkv, 0)
        out = R.run(lambda r, env: getattr(attns[r], fn)(sh(x, r), pos, cos_sin, kvs[r], 0))
        for kv in kvs.values():
            kv.kv_offset.fill_(S)
        out1 = R.run(lambda r, env: getattr(attns[r], fn)(sh(x1, r), pos1, cos_sin, kvs[r], 0))
        for r in R.local_ranks:
            assert rel_err(out[r], sh(ref, r)) < 3e-2, f"rank {r} prefill: {rel_err(out[r], sh(ref, r))}"
            assert rel_err(out1[r], sh(ref1, r)) < 3e-2, f"rank {r} decode: {rel_err(out1[r], sh(ref1, r))}"
        R.print(f"✅ TP_Attn {args.model} bsz={B} seq={S} mode={args.mode} world={w} check passed")
    else:
        ms = R.time(lambda r, env: getattr(attns[r], fn)(sh(x, r), pos, cos_sin, kvs[r], 0), args.warmup, args.iters)
        R.print_all(f"dist-triton attn prefill ({args.mode}): {ms:.4f} ms")
        if not R.virtual:
            r0 = R.local_ranks[0]
            ms_t = R.time(lambda r, env: attns[r0].torch_fwd(x, pos, cos_sin, kvs[r0], 0), args.warmup, args.iters)
            R.print_all(f"torch attn prefill: {ms_t:.4f} ms")
    R.close()


if __name__ == "__main__":
    main()
