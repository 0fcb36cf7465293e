"""Per-workgroup timeline of the MoE grouped GEMM (one rank of eight alone on the GPU, Mixtral shapes); see
trace_solo_rank.py for the method."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from triton_distributed_gh_amd import _lib  # noqa: E402
from triton_distributed_gh_amd.ops.ag_gemm import EPOCH_MOD  # noqa: E402
from triton_distributed_gh_amd.ops.moe import (ag_group_gemm, create_ag_group_gemm_context, create_moe_rs_context,  # noqa: E402
                                               run_moe_reduce_rs)
from triton_distributed_gh_amd.utils.testing import VirtualCluster  # noqa: E402
import trace_solo_rank as T  # noqa: E402

T.NAMES.update({3: "wait/sync", 6: "push"})
w, M, E, topk, hidden, ffn = 8, 16384, 8, 2, 4096, 14336
vc = VirtualCluster(w, heap_bytes=1536 << 20)
vc.envs[0].shared_gpu = False
dt = torch.bfloat16
g = torch.Generator(device="cpu").manual_seed(0)
ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(M)]).int().cuda()
lib = _lib.hip()
buf = torch.zeros(256 * T.CAP * 2, dtype=torch.int64, device="cuda")

ctxs = vc.collective(lambda r, env: create_ag_group_gemm_context(M, ffn // w, hidden, E, topk, dt, r, w, env=env))
c = ctxs[0]
A = (torch.rand(M // w, hidden, device="cuda") - 0.5).to(dt)
B = (torch.rand(E, ffn // w, hidden, device="cuda") - 0.5).to(dt)
rs = vc.collective(lambda r, env: create_moe_rs_context(r, w, w, M, hidden, E, topk, dt, env=env))
c2 = rs[0]
X = (torch.rand(M * topk, ffn // w, device="cuda") - 0.5).to(dt)
W2 = (torch.rand(E, hidden, ffn // w, device="cuda") - 0.5).to(dt)
rw = torch.rand(M, topk, device="cuda")


def run_ag():
    ag = c.ag
    ag.flags[ag.call_count & 1][0].fill_(ag.call_count % EPOCH_MOD + 1)
    return ag_group_gemm(A, B, c, ids)


def run_rs():
    c2.flags[c2.call_count & 1][0].fill_(c2.call_count % EPOCH_MOD + 1)
    return run_moe_reduce_rs(X, W2, ids, rw, c2)


for fn, name in ((run_ag, "ag_group_gemm"), (run_rs, "moe_reduce_rs")):
    buf.zero_()
    for _ in range(6):
        fn()
    _lib.check(lib.tdist_debug_set_wg_trace(ctypes.c_void_p(buf.data_ptr()), T.CAP), "set trace")
    fn()
    _lib.check(lib.tdist_debug_set_wg_trace(None, 0), "clear trace")
    torch.cuda.synchronize()
    T.analyse(buf, name)
vc.close()
