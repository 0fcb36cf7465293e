"""AOT kernel registry: list the registered kernels, then call the vector-add, matmul and
flash-decode entries through the reference's convention
``distributed.<kernel>(stream, args..., algo_info)`` and check them against PyTorch.

    python triton_dist/test/nvidia/test_compile_aot.py

(CLI of reference python/triton_dist/test/nvidia/test_compile_aot.py:162-236.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.tools import distributed, registered_kernels  # noqa: E402
from triton_distributed_gh_amd.tools import aot_kernels  # noqa: E402,F401  (registers the in-tree set)


def main():
    argparse.ArgumentParser().parse_args()
    names = registered_kernels()
    print(f"{len(names)} AOT kernels registered:")
    for n in names:
        print("   ", n)
    if not torch.cuda.is_available():
        print("no GPU: registry listing only")
        return
    s = torch.cuda.current_stream().cuda_stream
    a = torch.randn(1 << 20, device="cuda").bfloat16()
    b = torch.randn(1 << 20, device="cuda").bfloat16()
    y = torch.empty_like(a)
    info = distributed.test_aot_kernel_bf16__triton_algo_info_t(BLOCK_SIZE=1024, num_warps=4, num_stages=1)
    distributed.test_aot_kernel_bf16(s, a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), info)
    assert torch.equal(y, a + b)
    M, N, K = 1024, 1024, 512
    A = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
    B = (torch.rand(N, K, device="cuda") - 0.5).bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    mi = distributed.matmul_kernel_descriptor_persistent_bf16__triton_algo_info_t(
        BLOCK_SIZE_M=256, BLOCK_SIZE_N=256, BLOCK_SIZE_K=64, GROUP_SIZE_M=8, num_warps=8, num_stages=2)
    distributed.matmul_kernel_descriptor_persistent_bf16(s, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, mi)
    ref = A.float() @ B.float().t()
    assert (C.float() - ref).abs().max() / ref.abs().max() < 1e-2
    print("✅ AOT entries (vector add, persistent matmul) match PyTorch", flush=True)


if __name__ == "__main__":
    main()
