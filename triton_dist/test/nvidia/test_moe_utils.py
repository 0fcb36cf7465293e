"""MoE index utilities: block-aligned sort of (token, k) pairs by (expert, source rank) - the
gather index, per-block expert id and arrival segments the grouped GEMMs consume - GPU kernel vs
the CPU oracle and the native C++ oracle.

    python triton_dist/test/nvidia/test_moe_utils.py --ntokens 4096 --experts 32 --topk 4 --world 8

(CLI of reference python/triton_dist/test/nvidia/test_moe_utils.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.nvidia.moe_utils import _align_cpu, moe_align_block_size, moe_align_reference  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ntokens", type=int, default=2048)
    ap.add_argument("--experts", type=int, default=16)
    ap.add_argument("--topk", type=int, default=4)
    ap.add_argument("--world", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    g = torch.Generator(device="cpu").manual_seed(args.seed)
    T, E, K, w = args.ntokens, args.experts, args.topk, args.world
    ids = torch.stack([torch.randperm(E, generator=g)[:K] for _ in range(T)]).int()
    tpr = T // w
    for rank in range(w):
        got = moe_align_block_size(ids.to(dev), E, w, rank, tpr)
        cap = got.sorted_ids.numel()
        ref = _align_cpu(ids, E, w, rank, tpr, cap)
        nb = int(ref.meta[0])
        assert torch.equal(got.meta.cpu(), ref.meta)
        assert torch.equal(got.expert_of_block.cpu()[:nb], ref.expert_of_block[:nb])
        assert torch.equal(got.seg_range.cpu()[:nb], ref.seg_range[:nb])
        assert torch.equal(got.order.cpu()[:nb], ref.order[:nb])
        assert torch.equal(torch.sort(got.sorted_ids.cpu()).values, torch.sort(ref.sorted_ids).values)
        cpp = moe_align_reference(ids, E, w, rank, tpr)
        n = min(cpp.sorted_ids.numel(), cap)
        assert torch.equal(cpp.meta, ref.meta) and torch.equal(cpp.sorted_ids[:n], ref.sorted_ids[:n])
    print(f"✅ moe_align_block_size ntokens={T} experts={E} topk={K} world={w}: kernel == CPU oracle == C++ oracle",
          flush=True)


if __name__ == "__main__":
    main()
