"""What v_dot2_f32_bf16 computes (scripts/dev/bench_src/dot2_check.hip)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.tools.runtime import compile_hip, load_module
here = os.path.dirname(os.path.abspath(__file__))
mod = load_module(compile_hip(os.path.join(here, "bench_src", "dot2_check.hip")))
a = torch.randn(64, 2, device="cuda").bfloat16(); b = torch.randn(64, 2, device="cuda").bfloat16()
c = torch.zeros(64, device="cuda")
mod.get("dot2_check")((1,), (64,), None, a, b, c)
torch.cuda.synchronize()
ref = (a.float() * b.float()).sum(1) + 1.0
print("max err", (c - ref).abs().max().item(), c[:4].tolist(), ref[:4].tolist())
lo = a.float()[:, 0] * b.float()[:, 0] + 1.0
print("only element 0:", (c - lo).abs().max().item())
