"""Expert-parallel MoE layer end to end: router -> EP dispatch -> per-expert gated MLP on the rank
that owns the expert -> EP combine (weighted), against a dense PyTorch evaluation of the same layer.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_ep_moe_inference.py -M 1024 -N 2048 -G 8 --topk 4
    python triton_dist/test/nvidia/test_ep_moe_inference.py -M 64 -N 256 -I 128 -G 2 --topk 2 --virtual-ranks 4 --check

-M tokens per rank, -N hidden, -I expert intermediate size, -G experts per rank.  Expert weights are
generated from the seed on every rank, so each rank can evaluate the golden for its own tokens.
(Scenario of reference python/triton_dist/test/nvidia/test_ep_moe_inference.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from test_ep_a2a import dispatch_all  # noqa: E402  (same directory)
from triton_dist.layers.nvidia.ep_a2a_layer import EPAll2AllLayer  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402


def expert_mlp(x, w_gate_up, w_down):
    gu = x.float() @ w_gate_up.float().t()
    inter = gu.shape[1] // 2
    return ((torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]).to(x.dtype).float() @ w_down.float().t()).to(x.dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-M", type=int, default=256, help="tokens per rank")
    ap.add_argument("-N", type=int, default=1024, help="hidden size")
    ap.add_argument("-I", type=int, default=512, help="expert intermediate size")
    ap.add_argument("-G", type=int, default=4, help="experts per rank")
    ap.add_argument("--topk", type=int, default=2)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    assert R.device.type == "cuda", "EP dispatch / combine run on the GPU backend"
    w, dev, dt = R.world_size, R.device, torch.bfloat16
    T, H, I, topk, E = args.M, args.N, args.I, args.topk, w * args.G
    layers = R.collective(lambda r, env: EPAll2AllLayer(getattr(env, "tp_group", None), T, H, topk, r, E, w, w, dt,
                                                        env=env))
    g = torch.Generator(device="cpu").manual_seed(args.seed)
    w_gu = ((torch.rand(E, 2 * I, H, generator=g) - 0.5) * 0.1).to(dev).to(dt)
    w_dn = ((torch.rand(E, H, I, generator=g) - 0.5) * 0.1).to(dev).to(dt)
    xs, idx, wts = {}, {}, {}
    for r in R.local_ranks:
        gr = torch.Generator(device="cpu").manual_seed(args.seed + 1 + r)
        xs[r] = (torch.rand(T, H, generator=gr) - 0.5).to(dev).to(dt)
        logits = torch.rand(T, E, generator=gr)
        top = torch.topk(torch.softmax(logits, -1), topk, dim=-1)
        idx[r], wts[r] = top.indices.int().to(dev), (top.values / top.values.sum(-1, keepdim=True)).to(dev)

    def forward():
        recv = dispatch_all(R, layers, xs, idx)

        def experts(r, env):
            L = layers[r]
            out = torch.empty_like(recv[r])
            row = 0
            for le, cnt in enumerate(L.recv_per_expert.tolist()):
                e = r * args.G + le
                if cnt:
                    out[row:row + cnt] = expert_mlp(recv[r][row:row + cnt], w_gu[e], w_dn[e])
                row += cnt
            return L.combine(out, wts[r])

        return R.run(experts)

    outs = forward()
    if args.check:
        for r in R.local_ranks:
            layers[r].check_error()
            ref = torch.zeros(T, H, device=dev)
            for k in range(topk):
                for e in idx[r][:, k].unique().tolist():
                    m = idx[r][:, k] == e
                    ref[m] += wts[r][m, k:k + 1] * expert_mlp(xs[r][m], w_gu[e], w_dn[e]).float()
            assert rel_err(outs[r], ref) < 3e-2, f"rank {r}: {rel_err(outs[r], ref)}"
        R.print(f"✅ EP MoE layer tokens/rank={T} hidden={H} inter={I} experts={E} topk={topk} world={w} check passed")
    else:
        import time
        R.barrier()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            forward()
        R.barrier()
        R.print_all(f"EP MoE layer #ranks {w}: {(time.perf_counter() - t0) * 1e3 / args.iters:.3f} ms per forward")
    R.close()


if __name__ == "__main__":
    main()
