"""EP exchanges replayed from HIP graphs, 8 virtual ranks on one GPU (all ranks' kernels share the chip, so the
numbers are an upper bound for one rank of a real node): fast_all_to_all + compaction at the DeepSeek decode shape
(128 tokens x topk 8 x 7168 fp8 + scales) and the normal-mode dispatch -> combine round trip (bf16)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from triton_distributed_gh_amd.ops.all_to_all import (all_to_all_post_process, create_all_to_all_context, fast_all_to_all,
                                                      quantize_fp8_groupwise)
from triton_distributed_gh_amd.ops.ep_a2a import EPAll2AllLayer
from triton_distributed_gh_amd.utils.testing import VirtualCluster

w, G, T, topk, hidden = 8, 32, 128, 8, 7168
rows, E = T * topk, w * G
vc = VirtualCluster(w, heap_bytes=1 << 29)


def wall_us(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def capture(fn):
    gs = []
    for r in range(w):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=vc.streams[r]):
            fn(r, vc.envs[r])
        gs.append(g)
    return gs


ctxs = vc.collective(lambda r, env: create_all_to_all_context(rows, hidden, r, E, w, G, torch.float8_e4m3fn, env=env))
data = []
for r in range(w):
    g = torch.Generator(device="cpu").manual_seed(r)
    dest = torch.sort(torch.randint(0, E, (rows,), generator=g)).values
    cum = torch.zeros(E + 1, dtype=torch.int32)
    cum[1:] = torch.cumsum(torch.bincount(dest, minlength=E), 0)
    q, s = quantize_fp8_groupwise(torch.randn(rows, hidden, generator=g).cuda().bfloat16())
    data.append((q, cum.cuda(), s))


def a2a(r, env):
    fast_all_to_all(ctxs[r], *data[r])


def a2a_pp(r, env):
    sp, rv, sc = fast_all_to_all(ctxs[r], *data[r])
    all_to_all_post_process(ctxs[r], sp, rv, sc, padded=True)


for name, fn in (("fast_all_to_all", a2a), ("fast_all_to_all + compaction", a2a_pp)):
    vc.run(fn)
    torch.cuda.synchronize()
    eager = wall_us(lambda: vc.run(fn))
    gs = capture(fn)
    rep = wall_us(lambda: vc.run(lambda r, env: gs[r].replay()))
    print(f"{name}: all 8 ranks eager {eager:.1f} us, graph replay {rep:.1f} us per round", flush=True)
for c in ctxs:
    c.check_error()

layers = vc.collective(lambda r, env: EPAll2AllLayer(None, T, hidden, topk, r, E, w, w, torch.bfloat16, num_sm=16,
                                                     capacity_factor=2.0, env=env))
xs = [torch.randn(T, hidden, device="cuda").bfloat16() for _ in range(w)]
idx = [torch.stack([torch.randperm(E, device="cuda")[:topk] for _ in range(T)]).int() for _ in range(w)]
wts = [torch.rand(T, topk, device="cuda") for _ in range(w)]


def trip(r, env):
    L = layers[r]
    L.dispatch(xs[r], idx[r], padded=True)
    L.combine(L.expert_out_buffer(), wts[r])


vc.run(trip)
torch.cuda.synchronize()
eager = wall_us(lambda: vc.run(trip))
gs = capture(trip)
rep = wall_us(lambda: vc.run(lambda r, env: gs[r].replay()))
print(f"EP dispatch + combine (3 kernels + 1 index copy per rank): all 8 ranks eager {eager:.1f} us, graph replay {rep:.1f} us per round",
      flush=True)
for L in layers:
    L.check_error()
vc.close()
