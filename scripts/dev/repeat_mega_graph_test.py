"""Runs the body of tests/test_mega.py::test_graph_backend_matches_the_interpreter_on_a_small_graph N times and prints
the relative error of every compared tensor (an intermittent mismatch names its back-end and tensor this way)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.mega.model_builder import ModelBuilder
from triton_distributed_gh_amd.ops.elementwise import rms_norm  # noqa: F401
BF = torch.bfloat16
M, H, N, I = 5, 1024, 768, 512
torch.manual_seed(3)
x, r = torch.randn(M, H, device="cuda").to(BF), torch.randn(M, H, device="cuda").to(BF)
nw = (1 + 0.1 * torch.randn(H, device="cuda")).to(BF)
w1 = (torch.randn(N, H, device="cuda") / 32).to(BF)
wgu = (torch.randn(2 * I, H, device="cuda") / 32).to(BF)
w2 = (torch.randn(H, I, device="cuda") / 32).to(BF)
# fp32 reference
h = (x.float() + r.float()).to(BF).float()
def norm(t): return (t * torch.rsqrt((t * t).mean(-1, keepdim=True) + 1e-6)) * nw.float()
y1_ref = norm(x.float()).to(BF).float() @ w1.float().t()
gu = norm(h).to(BF).float() @ wgu.float().t()
act = (torch.nn.functional.silu(gu[:, :I].to(BF).float()) * gu[:, I:].to(BF).float()).to(BF).float()
y2 = act @ w2.float().t() + h
s_ref = y2.to(BF).float() + h
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    out = {}
    for backend in ("interpreter", "graph"):
        new = lambda *s: torch.zeros(*s, device="cuda", dtype=BF)   # noqa: E731
        xn, y1, hid, a_, y2_, s_ = new(M, H), new(M, N), new(M, H), new(M, I), new(M, H), new(M, H)
        b = ModelBuilder(num_queues=32)
        b.make_rms_norm(x, nw, xn, 1e-6)
        b.make_linear(xn, w1, y1)
        b.make_rms_norm(x, nw, xn, 1e-6, residual=r, residual_out=hid)
        b.make_fc1(xn, wgu, a_)
        b.make_fc2(a_, w2, y2_, residual=hid)
        b.make_add(y2_, hid, s_)
        b.compile(backend=backend)
        errs = []
        for rep in range(3):
            b.run()
            torch.cuda.synchronize()
            e = lambda t, ref: ((t.float() - ref).abs().max() / ref.abs().max()).item()   # noqa: E731
            errs.append((e(y1, y1_ref), e(hid, h), e(s_, s_ref)))
        b.check_error()
        out[backend] = errs
    bad = any(v > 2e-2 for b_ in out.values() for t in b_ for v in t)
    print(it, "BAD" if bad else "ok", {k: [tuple(round(v, 4) for v in t) for t in vs] for k, vs in out.items()}, flush=True)
