"""Per-workgroup timeline of the dense persistent GEMM at a given shape (same phases as trace_moe_solo.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from triton_distributed_gh_amd import _lib
from triton_distributed_gh_amd.ops import gemm as G
import trace_solo_rank as T

M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32768, 4096, 1792)))
a = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
b = (torch.rand(N, K, device="cuda") - 0.5).bfloat16()
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
lib = _lib.hip()
buf = torch.zeros(256 * T.CAP * 2, dtype=torch.int64, device="cuda")
for _ in range(5):
    G.gemm_nt(a, b, o, force_native=True)
_lib.check(lib.tdist_debug_set_wg_trace(ctypes.c_void_p(buf.data_ptr()), T.CAP), "set trace")
G.gemm_nt(a, b, o, force_native=True)
_lib.check(lib.tdist_debug_set_wg_trace(None, 0), "clear trace")
torch.cuda.synchronize()
T.analyse(buf, f"dense gemm {M}x{N}x{K}")
