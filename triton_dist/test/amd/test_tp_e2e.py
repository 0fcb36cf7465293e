"""Whole-model prefill / decode under tensor parallelism: --check against the torch forward, or timing.

    bash scripts/launch_amd.sh python/triton_dist/test/amd/test_tp_e2e.py --bsz 32 --seq_len 128 --model Qwen/Qwen3-32B --check
    python triton_dist/test/amd/test_tp_e2e.py --model tdist/tiny-dense --bsz 4 --seq_len 32 --virtual-ranks 2 --check

(CLI of reference python/triton_dist/test/amd/test_tp_e2e.py:80-91.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
os.environ.setdefault("RANDOM_PARAMS", "1")

import torch  # noqa: E402

from triton_dist.models import AutoLLM, KV_Cache, ModelConfig  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bsz", type=int, default=32)
    ap.add_argument("--seq_len", type=int, default=128)
    ap.add_argument("--model", default="Qwen/Qwen3-32B")
    ap.add_argument("--num_layers", type=int, default=None, help="truncate the stack (fits a smoke run on one GPU)")
    ap.add_argument("--mode", default="triton_dist", choices=["triton_dist", "triton_dist_AR", "triton_dist_gemm_ar"])
    ap.add_argument("--decode", action="store_true", help="time / check a decode step instead of the prefill")
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, dt = R.world_size, R.device, torch.bfloat16
    L = args.seq_len + 16
    models = {r: AutoLLM.from_pretrained(
        ModelConfig(model_name=args.model, max_length=L, dtype=dt, rank=r, world_size=w, num_layers=args.num_layers),
        R.envs[0].tp_group if not R.virtual else None, env=R.envs[r] if R.virtual else R.envs[0]) for r in R.local_ranks}
    m0 = next(iter(models.values()))
    kvs = {r: KV_Cache(m0.num_layers, args.bsz, L, m0.num_key_value_heads, m0.head_dim, dt, w, device=dev) for r in R.local_ranks}
    torch.manual_seed(args.seed)
    S = 1 if args.decode else args.seq_len
    ids = torch.randint(0, 1000, (args.bsz, S), device=dev)
    base = args.seq_len if args.decode else 0
    pos = (torch.arange(S, device=dev) + base)[None].repeat(args.bsz, 1)
    if args.decode:
        for kv in kvs.values():
            kv.rand_fill_kv_cache(args.seq_len)
            kv.kv_offset.fill_(args.seq_len)
    M = args.bsz * S

    def init(r, env):
        {"triton_dist": models[r].init_triton_dist_ctx, "triton_dist_AR": models[r].init_triton_dist_AR_ctx,
         "triton_dist_gemm_ar": models[r].init_triton_dist_gemm_ar_ctx}[args.mode](max_M=M)

    R.collective(init)
    per = args.bsz // w
    sharded = args.mode == "triton_dist"

    def fwd(r, env, mode):
        models[r].set_fwd(mode)
        x = ids[r * per:(r + 1) * per] if (sharded and mode != "torch") else ids
        return models[r].inference(x, pos, kvs[r])

    if args.check:
        assert not R.virtual, "--check compares with the RCCL torch forward: run under torchrun (or use tests/)"
        ref = R.run(lambda r, e: fwd(r, e, "torch"))
        out = R.run(lambda r, e: fwd(r, e, args.mode))
        for r in R.local_ranks:
            want = ref[r][r * per:(r + 1) * per] if sharded else ref[r]
            assert rel_err(out[r], want) < 5e-2, f"rank {r}: {rel_err(out[r], want)}"
        R.print(f"✅ e2e {args.model} bsz={args.bsz} seq={S} mode={args.mode} world={w} check passed")
    else:
        t = R.time(lambda r, e: fwd(r, e, args.mode), args.warmup, args.iters)
        R.print_all(f"dist-triton e2e ({args.mode}, {'decode' if args.decode else 'prefill'}): {t:.4f} ms")
        if not R.virtual:
            t0 = R.time(lambda r, e: fwd(r, e, "torch"), args.warmup, args.iters)
            R.print_all(f"torch e2e: {t0:.4f} ms  (speed-up {t0 / t:.3f}x)")
    R.close()


if __name__ == "__main__":
    main()
