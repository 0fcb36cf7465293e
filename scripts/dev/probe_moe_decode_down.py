"""MoE decode down projection alone (Qwen3-30B-A3B shapes: 128 experts, top-8, H 2048, Ip 768), 1 - 4 tokens, cold weights.

The form is chosen per process: TDIST_MOE_DOWN_NARROW=0 (16 columns per workgroup, MFMA) / 1 (8 columns, vector ALUs)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd import _lib
from triton_distributed_gh_amd.utils.perf import graph_bench

E, H, Ip, topk = 128, 2048, 768, 8
dev, bf = "cuda", torch.bfloat16
lib = _lib.hip()
FORM = int(os.environ.get("TDIST_MOE_DOWN_NARROW", "-1"))
down = [((torch.rand(E, H, Ip, device=dev) * 2 - 1) * 0.05).to(bf) for _ in range(2)]      # 2 x 403 MB
for n in (1, 2, 4, 8, 16):
    hbuf = (torch.rand(n * topk, Ip, device=dev) * 2 - 1).to(bf)
    wts = torch.full((n, topk), 1.0 / topk, device=dev)
    out = torch.empty(n, H, device=dev, dtype=bf)
    sets = [torch.stack([torch.randperm(E, device=dev)[:topk] for _ in range(n)]).int() for _ in range(40)]
    it = [0]

    def call():
        i = it[0]
        it[0] += 1
        _lib.check(lib.tdist_moe_decode_down(_lib.ptr(hbuf), _lib.ptr(sets[i % 40]), _lib.ptr(wts), _lib.ptr(down[i & 1]),
                                             _lib.ptr(out), n, topk, H, Ip, E, FORM, _lib.stream_handle()), "down")

    t = graph_bench(call, n=40)
    mb = n * topk * H * Ip * 2 / 1e6
    print(f"tokens {n}: {t:.2f} us for {mb:.0f} MB ({mb / t / 1e6:.2f} TB/s)  narrow={os.environ.get('TDIST_MOE_DOWN_NARROW', 'auto')}", flush=True)
