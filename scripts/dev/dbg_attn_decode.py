import os, sys, torch
os.environ["GPU_MAX_HW_QUEUES"] = "24"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from types import SimpleNamespace
from triton_distributed_gh_amd.layers import TP_Attn
from triton_distributed_gh_amd.layers.tp_attn import _set_cos_sin_cache
from triton_distributed_gh_amd.models import KV_Cache
from triton_distributed_gh_amd.ops.ag_gemm import ag_gemm_intra_node
from triton_distributed_gh_amd.ops.gemm_rs import gemm_rs_intra_node
from triton_distributed_gh_amd.utils.testing import VirtualCluster
import test_layers_gpu as T
BF = torch.bfloat16
w, B, S, H, Hq, Hkv, D = 4, 8, 32, 512, 8, 8, 64
torch.manual_seed(2)
tree = T._attn_tree(H, Hq, Hkv, D)
inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device="cuda").float() / D))
cos_sin = _set_cos_sin_cache(inv, 256)
vc = VirtualCluster(w, heap_bytes=1 << 28)
attns = [TP_Attn(rank=r, world_size=w, env=vc.envs[r]) for r in range(w)]
for a in attns: a._init_parameters(tree)
vc.collective(lambda r, env: attns[r]._init_ctx(max_M=B * S))
per = B // w
for trial in range(3):
    x1 = (torch.rand(B, 1, H, device="cuda") * 2 - 1).to(BF)
    torch.cuda.synchronize()
    qkv = vc.run(lambda r, env: ag_gemm_intra_node(x1[r*per:(r+1)*per].reshape(-1, H), attns[r].wqkv, False, attns[r].ag_ctx))
    for r in range(w):
        ref = x1.reshape(B, H).float() @ attns[r].wqkv.float().t()
        print("trial", trial, "rank", r, "ag_gemm rel", T._rel(qkv[r], ref), flush=True)
    o = [(torch.rand(B, Hq * D // w, device="cuda") * 2 - 1).to(BF) for r in range(w)]
    torch.cuda.synchronize()
    outs = vc.run(lambda r, env: gemm_rs_intra_node(o[r], attns[r].wo, attns[r].rs_ctx))
    full = sum((o[r].float() @ attns[r].wo.float().t()) for r in range(w))
    for r in range(w):
        print("trial", trial, "rank", r, "gemm_rs rel", T._rel(outs[r], full[r*per:(r+1)*per]), flush=True)
# attention decode alone
from triton_distributed_gh_amd.ops.flash_attn import flash_attn_with_kvcache, attention_reference
Hl = Hq // w
kc = torch.randn(B, 256, Hl, D, device="cuda").to(BF); vcache = torch.randn(B, 256, Hl, D, device="cuda").to(BF)
q = torch.randn(B, 1, Hl, D, device="cuda").to(BF); k = torch.randn(B, 1, Hl, D, device="cuda").to(BF); v = torch.randn(B, 1, Hl, D, device="cuda").to(BF)
lens = torch.full((B,), S, device="cuda", dtype=torch.int32)
out = flash_attn_with_kvcache(q, kc, vcache, k, v, cache_seqlens=lens)
for b in range(B):
    ref = attention_reference(q[b], kc[b, :S+1], vcache[b, :S+1], [0, 1], [0, S+1], True)
    print("decode attn b", b, T._rel(out[b], ref), flush=True)
