"""A few batch-1 decode attention calls with cold K / V for a kernel trace: L and the split count from argv."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.ops import flash_decode

L, sp = int(sys.argv[1]), int(sys.argv[2])
dev, bf, Hq, Hkv, D, B = "cuda", torch.bfloat16, 32, 8, 128, 1
copies = max(3, int(1200e6 / (2 * L * Hkv * D * 2)) + 1)
kcs = [torch.randn(B, L, Hkv, D, device=dev).to(bf) for _ in range(copies)]
vcs = [torch.randn(B, L, Hkv, D, device=dev).to(bf) for _ in range(copies)]
q = torch.randn(B, Hq, D, device=dev).to(bf)
lens = torch.full((B,), L, dtype=torch.int32, device=dev)
table = torch.arange(B, dtype=torch.int32, device=dev).view(B, 1)
for i in range(3 * copies):
    flash_decode.gqa_fwd_batch_decode(q, kcs[i % copies], vcs[i % copies], None, None, lens, table, D ** -0.5, kv_split=sp)
torch.cuda.synchronize()
