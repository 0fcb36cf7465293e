"""Compute-side cost of AG+MoE grouped GEMM and MoE+RS for ONE rank at the Mixtral-8x7B shapes:
rank 0 of an 8-rank virtual cluster runs alone on the whole GPU with the arrival flags pre-set
(nothing waits, every push lands in local HBM).  Run under rocprofv3 --kernel-trace for the
per-kernel split; prints HIP-event times per call."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from triton_distributed_gh_amd.ops.ag_gemm import EPOCH_MOD
from triton_distributed_gh_amd.ops.moe import (ag_group_gemm, create_ag_group_gemm_context, create_moe_rs_context,
                                               run_moe_reduce_rs)
from triton_distributed_gh_amd.utils.testing import VirtualCluster

w, M, E, topk, hidden, ffn = 8, 16384, 8, 2, 4096, 14336
iters = int(os.environ.get("ITERS", "12"))
vc = VirtualCluster(w, heap_bytes=1536 << 20)
env0 = vc.envs[0]
env0.shared_gpu = False      # rank 0 owns the GPU in this probe: full-size grids
dt = torch.bfloat16
g = torch.Generator(device="cpu").manual_seed(0)
ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(M)]).int().cuda()


def timed(fn, prep):
    ts = []
    for _ in range(iters):
        prep()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2], ts[0]


def burst(fn, prep, n=10):
    """n calls enqueued back to back (what a model's layer loop does): time per call without the host's share."""
    for _ in range(2):
        prep(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        prep(); fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


ONLY = os.environ.get("ONLY", "")          # "ag" / "rs": one of the two ops (per-op counter runs)

# ---- AG + grouped GEMM (w1: N = ffn, K = hidden) ----
ctxs = vc.collective(lambda r, env: create_ag_group_gemm_context(M, ffn // w, hidden, E, topk, dt, r, w, env=env))
c = ctxs[0]
A = (torch.rand(M // w, hidden, device="cuda") - 0.5).to(dt)
B = (torch.rand(E, ffn // w, hidden, device="cuda") - 0.5).to(dt)

def prep_ag():
    ag = c.ag
    ag.flags[ag.call_count & 1][0].fill_(ag.call_count % EPOCH_MOD + 1)

fl = 2.0 * M * topk * (ffn // w) * hidden
if ONLY in ("", "ag"):
    med, mn = timed(lambda: ag_group_gemm(A, B, c, ids), prep_ag)
    print(f"ag_group_gemm  solo rank: median {med:.1f} us  min {mn:.1f} us  ({fl / med / 1e6:.0f} TFLOP/s); "
          f"back to back {burst(lambda: ag_group_gemm(A, B, c, ids), prep_ag):.1f} us per call", flush=True)
    c.check_error()
if ONLY == "ag":
    vc.close()
    sys.exit(0)

# ---- grouped GEMM + topk reduce + RS (w2: N = hidden, K = ffn / w) ----
rs = vc.collective(lambda r, env: create_moe_rs_context(r, w, w, M, hidden, E, topk, dt, env=env))
c2 = rs[0]
X = (torch.rand(M * topk, ffn // w, device="cuda") - 0.5).to(dt)
W2 = (torch.rand(E, hidden, ffn // w, device="cuda") - 0.5).to(dt)
rw = torch.rand(M, topk, device="cuda")

def prep_rs():
    c2.flags[c2.call_count & 1][0].fill_(c2.call_count % EPOCH_MOD + 1)

fl = 2.0 * M * topk * hidden * (ffn // w)
from triton_distributed_gh_amd import autotuner
sweep = [(1, 0), (2, 16), (2, 32), (4, 16), (4, 32), (4, 48), (8, 32)]
if os.environ.get("NCK"):
    sweep = [(int(os.environ["NCK"]), int(os.environ.get("RS_CUS", "32")))]
for nck, rs_cus in sweep:
    if rs_cus:
        autotuner._ACTIVE["moe_rs.RS_CUS"] = rs_cus
    med, mn = timed(lambda: run_moe_reduce_rs(X, W2, ids, rw, c2, n_chunks=nck), prep_rs)
    bb = burst(lambda: run_moe_reduce_rs(X, W2, ids, rw, c2, n_chunks=nck), prep_rs)
    print(f"moe_reduce_rs  solo rank, n_chunks {nck} (reduce keeps {rs_cus} CUs): median {med:.1f} us  min {mn:.1f} us  "
          f"({fl / med / 1e6:.0f} TFLOP/s); back to back {bb:.1f} us per call", flush=True)
    c2.check_error()
vc.close()
