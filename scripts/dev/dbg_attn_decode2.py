import os, sys, torch
os.environ["GPU_MAX_HW_QUEUES"] = "24"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from triton_distributed_gh_amd.layers import TP_Attn
from triton_distributed_gh_amd.layers.tp_attn import _set_cos_sin_cache
from triton_distributed_gh_amd.models import KV_Cache
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
rec = {}
for r, a in enumerate(attns):
    orig = a._attend
    def wrapped(qkv, position_ids, csc, kv, li, _o=orig, _r=r):
        out = _o(qkv, position_ids, csc, kv, li)
        rec[_r] = (qkv.clone(), out.clone())
        return out
    a._attend = wrapped
kvs = [KV_Cache(1, B, 256, Hkv, D, BF, w) for _ in range(w)]
x = (torch.rand(B, S, H, device="cuda") * 2 - 1).to(BF)
pos = torch.arange(S, device="cuda")[None].repeat(B, 1)
ref, pk, pv = T._attn_ref(x, tree, pos, cos_sin)
torch.cuda.synchronize()
outs = vc.run(lambda r, env: attns[r].dist_triton_fwd(x[r*per:(r+1)*per], pos, cos_sin, kvs[r], 0))
print("prefill rel", T._rel(torch.cat(outs), ref))
for kv in kvs: kv.kv_offset.fill_(S)
x1 = (torch.rand(B, 1, H, device="cuda") * 2 - 1).to(BF)
pos1 = torch.full((B, 1), S, device="cuda", dtype=torch.long)
ref1, k1, v1 = T._attn_ref(x1, tree, pos1, cos_sin, pk, pv)
torch.cuda.synchronize()
outs = vc.run(lambda r, env: attns[r].dist_triton_fwd(x1[r*per:(r+1)*per], pos1, cos_sin, kvs[r], 0))
print("decode rel", T._rel(torch.cat(outs), ref1))
for r in range(w):
    qkv, out = rec[r]
    refq = x1.reshape(B, H).float() @ attns[r].wqkv.float().t()
    print("rank", r, "qkv rel", T._rel(qkv.reshape(B, -1), refq))
    # check cache contents vs reference k (post norm+rope) for this rank's heads
    hl = Hkv // w
    kc = kvs[r].k_cache[0][:, :S + 1]
    print("   kcache rel", T._rel(kc, k1[:, :, r*hl:(r+1)*hl]), " vcache rel", T._rel(kvs[r].v_cache[0][:, :S+1], v1[:, :, r*hl:(r+1)*hl]))
    # attention out reference from ref q: recompute
    full_o = None
# reference attention output before o_proj
xf = x1.float()
q = (xf @ tree.q_proj.weight.float().t()).view(B, 1, -1, D)
def rms(z, n): return z * torch.rsqrt(z.pow(2).mean(-1, keepdim=True) + 1e-6) * n.weight.float()
c = cos_sin[0][pos1].unsqueeze(2); s = cos_sin[1][pos1].unsqueeze(2)
q = rms(q, tree.q_norm); q = torch.cat([q[..., :D//2]*c - q[..., D//2:]*s, q[..., D//2:]*c + q[..., :D//2]*s], -1)
sc = torch.einsum("bqhd,bkhd->bhqk", q, k1) / D ** 0.5
p = torch.softmax(sc, -1)
o = torch.einsum("bhqk,bkhd->bqhd", p, v1)
hl = Hq // w
for r in range(w):
    print("rank", r, "attn-out rel", T._rel(rec[r][1].reshape(B, hl, D), o[:, 0, r*hl:(r+1)*hl]))
    print("   final partial check: rs input finite", torch.isfinite(rec[r][1].float()).all().item())
