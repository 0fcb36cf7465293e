import os, sys, torch
os.environ["GPU_MAX_HW_QUEUES"] = "24"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from triton_distributed_gh_amd.layers import TP_Attn
import triton_distributed_gh_amd.layers.tp_attn as TA
from triton_distributed_gh_amd.layers.tp_attn import _set_cos_sin_cache
from triton_distributed_gh_amd.models import KV_Cache
from triton_distributed_gh_amd.ops.flash_attn import attention_reference
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
rec = []
orig = TA.flash_attn_with_kvcache
def wrapped(q, k_cache, v_cache, k=None, v=None, cache_seqlens=None, **kw):
    out = orig(q=q, k_cache=k_cache, v_cache=v_cache, k=k, v=v, cache_seqlens=cache_seqlens, **kw)
    rec.append((q.clone(), k.clone(), v.clone(), cache_seqlens.clone(), k_cache, v_cache, out.clone()))
    return out
TA.flash_attn_with_kvcache = wrapped
for trial in range(3):
    rec.clear()
    kvs = [KV_Cache(1, B, 256, Hkv, D, BF, w) for _ in range(w)]
    x = (torch.rand(B, S, H, device="cuda") * 2 - 1).to(BF)
    pos = torch.arange(S, device="cuda")[None].repeat(B, 1)
    torch.cuda.synchronize()
    outs = vc.run(lambda r, env: attns[r].dist_triton_fwd(x[r*per:(r+1)*per], pos, cos_sin, kvs[r], 0))
    for kv in kvs: kv.kv_offset.fill_(S)
    x1 = (torch.rand(B, 1, H, device="cuda") * 2 - 1).to(BF)
    pos1 = torch.full((B, 1), S, device="cuda", dtype=torch.long)
    torch.cuda.synchronize()
    rec.clear()
    outs = vc.run(lambda r, env: attns[r].dist_triton_fwd(x1[r*per:(r+1)*per], pos1, cos_sin, kvs[r], 0))
    for r, (q, k, v, lens, kc, vcache, out) in enumerate(rec):
        errs = []
        for b in range(B):
            L = int(lens[b]) + 1
            ref = attention_reference(q[b], kc[b, :L], vcache[b, :L], [0, 1], [0, L], True)
            errs.append(T._rel(out[b], ref))
        kerr = T._rel(kc[:, S], k[:, 0]); verr = T._rel(vcache[:, S], v[:, 0])
        print("trial", trial, "rank", r, "lens", lens.tolist(), "max err %.4f" % max(errs), "append err", kerr, verr, flush=True)
