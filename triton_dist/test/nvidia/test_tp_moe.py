"""TP MoE layer (AG + grouped GEMM -> SiLU*up -> grouped GEMM + top-k reduce + RS) vs the torch
per-expert loop.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_tp_moe.py --M 4096 --model Qwen/Qwen3-30B-A3B
    python triton_dist/test/nvidia/test_tp_moe.py --M 64 --model tdist/tiny-moe --virtual-ranks 2 --check

(CLI of reference python/triton_dist/test/nvidia/test_tp_moe.py; layer 0 of the model, random weights.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
os.environ.setdefault("RANDOM_PARAMS", "1")

import torch  # noqa: E402

from triton_dist.layers.nvidia.tp_moe import TP_MoE  # noqa: E402
from triton_distributed_gh_amd.models.config import ModelConfig, load_arch_config  # noqa: E402
from triton_distributed_gh_amd.models.weights import RandomWeights  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=4096, help="tokens over all ranks")
    ap.add_argument("--model", default="Qwen/Qwen3-30B-A3B")
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, M = R.world_size, R.device, args.M
    dt = torch.bfloat16 if dev.type == "cuda" else torch.float32
    arch = load_arch_config(ModelConfig(model_name=args.model))
    tree = RandomWeights(arch, dt, dev).make_layer(0).mlp
    moes = {r: TP_MoE(rank=r, world_size=w, group=None if R.virtual else R.envs[0].tp_group,
                      env=R.envs[r] if R.virtual else R.envs[0]) for r in R.local_ranks}
    for m in moes.values():
        m._init_parameters(tree)
    R.collective(lambda r, env: moes[r]._init_ctx(M))
    torch.manual_seed(args.seed)
    x = (torch.rand(M, arch.hidden_size, device=dev) * 2 - 1).to(dt)
    per = M // w
    fwd = lambda r, env: moes[r].dist_triton_fwd(x[r * per:(r + 1) * per])   # noqa: E731
    if args.check:
        gold = TP_MoE(rank=0, world_size=1, env=R.envs[R.local_ranks[0]] if R.virtual else R.envs[0])
        gold._init_parameters(tree)
        ref = gold.torch_fwd(x)
        for it in range(2):
            out = R.run(fwd)
            for r in R.local_ranks:
                err = rel_err(out[r], ref[r * per:(r + 1) * per])
                assert err < 4e-2, f"rank {r} iteration {it}: {err}"
        R.print(f"✅ TP_MoE {args.model} M={M} world={w} check passed")
    else:
        R.print_all(f"dist-triton moe: {R.time(fwd, args.warmup, args.iters):.4f} ms")
        if not R.virtual:
            r0 = R.local_ranks[0]
            R.print_all(f"torch moe: {R.time(lambda r, e: moes[r0].torch_fwd(x), args.warmup, args.iters):.4f} ms")
    R.close()


if __name__ == "__main__":
    main()
