"""TP MLP layer: torch forward vs AG+GEMM / GEMM+RS (and all-reduce, GEMM+AR) forwards.

    CUDA_GRAPH=1 bash scripts/launch_amd.sh python/triton_dist/test/amd/test_tp_mlp.py --M 4096 --model Qwen/Qwen3-32B
    ... --per_op          times AG-GEMM and GEMM-RS separately, as the reference's table does
    python triton_dist/test/amd/test_tp_mlp.py --M 256 --model tdist/tiny-dense --virtual-ranks 2 --check

(CLI of reference python/triton_dist/test/amd/test_tp_mlp.py:85-95; layer 0 of the model, random weights.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
os.environ.setdefault("RANDOM_PARAMS", "1")

import torch  # noqa: E402

from triton_dist.layers.amd.tp_mlp import TP_MLP  # noqa: E402
from triton_distributed_gh_amd.models.config import ModelConfig, load_arch_config  # noqa: E402
from triton_distributed_gh_amd.models.weights import RandomWeights  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--model", default="Qwen/Qwen3-32B")
    ap.add_argument("--per_op", action="store_true")
    ap.add_argument("--mode", default="ag_rs", choices=["ag_rs", "allreduce", "gemm_ar"])
    add_common_args(ap)
    ap.set_defaults(warmup=20, iters=100)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, dt, M = R.world_size, R.device, torch.bfloat16, args.M
    arch = load_arch_config(ModelConfig(model_name=args.model))
    tree = RandomWeights(arch, dt, dev).make_layer(0).mlp
    H = arch.hidden_size
    mlps = {r: TP_MLP(rank=r, world_size=w, group=R.envs[0].tp_group if not R.virtual else None,
                      env=R.envs[r] if R.virtual else R.envs[0]) for r in R.local_ranks}
    for m in mlps.values():
        m._init_parameters(tree)

    def init(r, env):
        mlps[r]._init_ctx(max_M=M)
        if args.mode == "allreduce":
            mlps[r]._init_AR_ctx(max_M=M)
        if args.mode == "gemm_ar":
            mlps[r]._init_gemm_ar_ctx(max_M=M)

    R.collective(init)
    torch.manual_seed(args.seed)
    x = (torch.rand(M, H, device=dev) * 2 - 1).to(dt)            # same on every rank (same seed)
    Mr = M // w
    fwd = {"ag_rs": lambda r: mlps[r].dist_triton_fwd(x[r * Mr:(r + 1) * Mr]),
           "allreduce": lambda r: mlps[r].dist_triton_AR_fwd(x), "gemm_ar": lambda r: mlps[r].dist_triton_gemm_ar_fwd(x)}[args.mode]
    if args.check:
        xf = x.float()
        g = torch.nn.functional.silu(xf @ tree.gate_proj.weight.float().t()) * (xf @ tree.up_proj.weight.float().t())
        ref = g @ tree.down_proj.weight.float().t()
        out = R.run(lambda r, env: fwd(r))
        for r in R.local_ranks:
            want = ref[r * Mr:(r + 1) * Mr] if args.mode == "ag_rs" else ref
            assert rel_err(out[r], want) < 3e-2, f"rank {r}: {rel_err(out[r], want)}"
        R.print(f"✅ TP_MLP {args.model} M={M} mode={args.mode} world={w} check passed")
    elif args.per_op:
        I = tree.gate_proj.weight.shape[0]
        y = (torch.rand(M, I // w, device=dev) * 2 - 1).to(dt)
        for name, fn in (("ag_gemm", lambda r, e: mlps[r].dist_triton_ag_gemm(x[r * Mr:(r + 1) * Mr])),
                         ("gemm_rs", lambda r, e: mlps[r].dist_triton_gemm_rs(y))):
            R.print_all(f"dist-triton {name}: {R.time(fn, args.warmup, args.iters):.4f} ms")
        if not R.virtual:
            for name, fn in (("ag_gemm", lambda r, e: mlps[r].torch_ag_gemm(x[r * Mr:(r + 1) * Mr])),
                             ("gemm_rs", lambda r, e: mlps[r].torch_gemm_rs(y))):
                R.print_all(f"torch {name}: {R.time(fn, args.warmup, args.iters):.4f} ms")
    else:
        run = lambda r, e: fwd(r)
        # CUDA_GRAPH=1 (reference: kernels/amd/all_gather_gemm.py:310, amd-ci.yml:48-52): one forward is
        # captured per process and replayed.  The fused kernels number their calls on the device, so a
        # replay is a fresh call; AG+GEMM / GEMM+RS accept capture for whole 256-row tiles per rank.
        capturable = args.mode == "allreduce" or (args.mode == "ag_rs" and Mr % 256 == 0) or w == 1
        if os.environ.get("CUDA_GRAPH", "0") == "1" and not R.virtual and dev.type == "cuda" and capturable:
            r0 = R.local_ranks[0]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    fwd(r0)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fwd(r0)
            run = lambda r, e: graph.replay()
            R.print("HIP graph captured")
        R.print_all(f"dist-triton mlp ({args.mode}): {R.time(run, args.warmup, args.iters):.4f} ms")
        if not R.virtual:
            R.print_all(f"torch mlp: {R.time(lambda r, e: mlps[r].torch_fwd(x), args.warmup, args.iters):.4f} ms")
    R.close()


if __name__ == "__main__":
    main()
