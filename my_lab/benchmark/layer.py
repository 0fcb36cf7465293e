"""bf16 linear layer ``y = x W^T + b`` on the gfx950 MFMA GEMM (role of the reference's
my_lab/benchmark/layer.py:20-64, an autotuned Triton matmul behind an nn.Module)."""
from __future__ import annotations

import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from triton_distributed_gh_amd.ops.gemm import gemm_nt  # noqa: E402


class NativeLinear(torch.nn.Module):
    """Drop-in for ``torch.nn.Linear`` (weight [out, in], K-contiguous - exactly the operand layout
    of ``gemm_nt``).  Shapes the kernel does not cover fall back to ``torch.matmul`` inside gemm_nt."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device="cuda", dtype=torch.bfloat16):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(out_features, in_features, device=device, dtype=dtype))
        self.bias = torch.nn.Parameter(torch.zeros(out_features, device=device, dtype=dtype)) if bias else None
        torch.nn.init.normal_(self.weight, std=in_features ** -0.5)

    @classmethod
    def from_torch(cls, lin: torch.nn.Linear) -> "NativeLinear":
        m = cls(lin.in_features, lin.out_features, lin.bias is not None, lin.weight.device, lin.weight.dtype)
        m.weight.data.copy_(lin.weight.data)
        if lin.bias is not None:
            m.bias.data.copy_(lin.bias.data)
        return m

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lead = x.shape[:-1]
        y = gemm_nt(x.reshape(-1, x.shape[-1]).contiguous(), self.weight.data,
                    bias=None if self.bias is None else self.bias.data)
        return y.view(*lead, -1)
