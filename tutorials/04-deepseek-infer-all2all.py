"""Tutorial 04 - DeepSeek-style expert-parallel inference step: fp8 quantise -> low-latency
all-to-all dispatch -> expert-side dequantise (the combine is the same call in reverse).

Shapes follow the reference's headline number (README.md:97: 128 tokens / rank, top-k 8, hidden 7168,
fp8, EP = world size).  Tokens are sorted by destination expert, quantised per 128-wide group
(scale = max|x| / 448, OCP e4m3), and each destination rank receives its rows, the per-expert
splits and the scales in ONE kernel with one flag (csrc/ops/comm.hip: all_to_all_kernel).

    python tutorials/04-deepseek-infer-all2all.py --virtual-ranks 4 --tokens 64 --hidden 1024
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402,F401

from triton_dist.kernels.nvidia import all_to_all_post_process, create_all_to_all_context, fast_all_to_all  # noqa: E402
from triton_distributed_gh_amd.ops.elementwise import fp8_group_quant  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=7168)
    ap.add_argument("--topk", type=int, default=8)
    ap.add_argument("--experts_per_rank", type=int, default=8)
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    assert R.device.type == "cuda", "fp8 quantisation runs on the GPU"
    w, dev, T, H, K, EPR = R.world_size, R.device, args.tokens, args.hidden, args.topk, args.experts_per_rank
    E = w * EPR
    max_m = T * K
    fp8 = torch.float8_e4m3fn
    ctxs = R.collective(lambda r, env: create_all_to_all_context(max_m, H, r, E, w, EPR, fp8, torch.float32, env=env))
    x = {r: torch.randn(T, H, device=dev).bfloat16() for r in R.local_ranks}
    ids = {r: torch.stack([torch.randperm(E, device=dev)[:K] for _ in range(T)]) for r in R.local_ranks}

    def step(r, env):
        flat = ids[r].reshape(-1)
        order = torch.sort(flat, stable=True).indices                    # (token, k) pairs sorted by expert
        rows = x[r][order // K].contiguous()
        q, scale = fp8_group_quant(rows)                                 # e4m3 + f32 scale per 128 columns
        cum = torch.zeros(E + 1, dtype=torch.int32, device=dev)
        cum[1:] = torch.cumsum(torch.bincount(flat, minlength=E), 0)
        splits, recv, rscale = fast_all_to_all(ctxs[r], q, cum, scale)   # dispatch
        toks, scales, per_exp = all_to_all_post_process(ctxs[r], splits, recv, rscale)
        # "expert compute": dequantise and scale by 2 so the round trip is checkable
        deq = (toks.float().view(toks.shape[0], H // 128, 128) * scales.unsqueeze(-1)).view(toks.shape[0], H)
        y = (deq * 2).bfloat16()
        return y, order, per_exp

    out = R.run(step)
    for r in R.local_ranks:
        y, order, per_exp = out[r]
        assert y.shape[0] == int(per_exp.sum()) and torch.isfinite(y.float()).all()
    R.print(f"✅ tutorial 04: dispatched {T * K} (token, expert) rows per rank in fp8 with scales over {w} ranks")
    R.close()


if __name__ == "__main__":
    main()
