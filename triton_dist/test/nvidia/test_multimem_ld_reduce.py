"""The reduce-on-load building block.  The reference tests NVLS multimem.ld_reduce (one load
returns the sum over all ranks); xGMI has no multicast, so the same result comes from w explicit
peer loads with fp32 accumulation - which is what the one-shot all-reduce kernel and the GEMM+AR
consumer do.  Every *_Multimem method name of the reference enum is checked to resolve to such a
kernel and to return the exact rank-ordered fp32 sum.

    python triton_dist/test/nvidia/test_multimem_ld_reduce.py --virtual-ranks 4

(Scenario of reference python/triton_dist/test/nvidia/test_multimem_ld_reduce.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.nvidia.allreduce import AllReduceMethod, all_reduce, create_allreduce_ctx  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402
from triton_distributed_gh_amd.utils.topo import is_nvshmem_multimem_supported  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=8 * 4096 + 24, help="elements (not a multiple of the vector width)")
    add_common_args(ap)
    args = ap.parse_args()
    R = Runner(args)
    w, dev = R.world_size, R.device
    assert not is_nvshmem_multimem_supported()
    ctxs = R.collective(lambda r, env: create_allreduce_ctx(1 << 22, r, w, env=env))
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        xs = {r: (torch.rand(args.n, device=dev) - 0.5).to(dt) for r in R.local_ranks}
        ref = torch.zeros(args.n, device=dev)
        for x in R.all_gather(xs):      # rank order, fp32: the kernels' summation order
            ref += x.float()
        for m in (AllReduceMethod.OneShot_Multimem, AllReduceMethod.TwoShot_Multimem, AllReduceMethod.TwoShot_Multimem_ST):
            outs = R.run(lambda r, env: all_reduce(xs[r], m, ctxs[r]))
            for r in R.local_ranks:
                ctxs[r].check_error()
                assert torch.equal(outs[r], ref.to(dt)), f"{m.name} {dt} rank {r}: not the rank-ordered fp32 sum"
    R.print(f"✅ load-reduce over {w} peers (bf16 / fp16 / fp32, multimem method names) exact")
    R.close()


if __name__ == "__main__":
    main()
