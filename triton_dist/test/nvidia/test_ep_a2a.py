"""Expert-parallel dispatch / combine (DeepEP-style normal mode) through EPAll2AllLayer.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_ep_a2a.py -M 4096 -N 7168 --topk 8 -G 32
    python triton_dist/test/nvidia/test_ep_a2a.py -M 256 -N 1024 --topk 4 -G 2 --virtual-ranks 4 --check

--check: "expert compute" multiplies each received row by (global expert id + 1); the combined
result must equal sum_k weight_k * x * (expert_k + 1) on the source rank.
(CLI of reference python/triton_dist/test/nvidia/test_ep_a2a.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import threading  # noqa: E402

from triton_dist.layers.nvidia.ep_a2a_layer import EPAll2AllLayer  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def dispatch_all(R, layers, xs, idx):
    """dispatch() reads its receive count on the host, which blocks until every rank has launched
    its count exchange: with virtual ranks the ranks must therefore dispatch from separate threads."""
    if not R.virtual:
        r = R.local_ranks[0]
        return {r: layers[r].dispatch(xs[r], idx[r])}
    outs, errs = {}, []

    def work(r):
        try:
            with torch.cuda.stream(R.vc.streams[r]):
                outs[r] = layers[r].dispatch(xs[r], idx[r])
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(r,)) for r in R.local_ranks]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if errs:
        raise errs[0]
    torch.cuda.synchronize()
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-M", type=int, default=4096, help="tokens per rank")
    ap.add_argument("-N", type=int, default=7168, help="hidden size")
    ap.add_argument("-G", type=int, default=32, help="experts per rank")
    ap.add_argument("--topk", type=int, default=8)
    ap.add_argument("--sm_margin", "--num-sm", dest="num_sm", type=int, default=16)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev = R.world_size, R.device
    T, H, topk, E = args.M, args.N, args.topk, w * args.G
    layers = R.collective(lambda r, env: EPAll2AllLayer(getattr(env, "tp_group", None), T, H, topk, r, E, w, w,
                                                        torch.bfloat16, num_sm=args.num_sm,
                                                        capacity_factor=float(w), env=env))
    xs = {r: torch.randn(T, H, device=dev).bfloat16() for r in R.local_ranks}
    idx = {r: torch.stack([torch.randperm(E, device=dev)[:topk] for _ in range(T)]).int() for r in R.local_ranks}
    wts = {r: torch.rand(T, topk, device=dev) for r in R.local_ranks}
    R.barrier()
    if args.check:
        recv = dispatch_all(R, layers, xs, idx)
        outs_e = {}
        for r in R.local_ranks:
            layers[r].check_error()
            rows, o = recv[r].clone(), 0
            for e_local, cnt in enumerate(layers[r].recv_per_expert.tolist()):
                rows[o:o + cnt] = (rows[o:o + cnt].float() * (r * args.G + e_local + 1)).bfloat16()
                o += cnt
            outs_e[r] = rows
        R.barrier()
        outs = R.run(lambda r, env: layers[r].combine(outs_e[r], wts[r]))
        for r in R.local_ranks:
            layers[r].check_error()
            ref = torch.zeros(T, H, device=dev)
            for k in range(topk):
                ref += wts[r][:, k:k + 1] * (xs[r].float() * (idx[r][:, k].long() + 1).unsqueeze(1)).bfloat16().float()
            assert torch.allclose(outs[r].float(), ref, rtol=2e-2, atol=2e-2 * topk), f"rank {r}"
        R.print(f"✅ ep dispatch/combine tokens={T} hidden={H} topk={topk} experts={E} world={w} check passed")
    else:
        import time
        for phase in range(args.warmup + args.iters):
            if phase == args.warmup:
                R.barrier()
                t0 = time.perf_counter()
            recv = dispatch_all(R, layers, xs, idx)
            R.run(lambda r, env: layers[r].combine(recv[r], wts[r]))
        R.barrier()
        ms = (time.perf_counter() - t0) * 1e3 / args.iters
        R.print_all(f"ep dispatch+combine #ranks {w}: {ms:.3f} ms per round trip ({T} tokens x topk {topk} x {H})")
    R.close()


if __name__ == "__main__":
    main()
