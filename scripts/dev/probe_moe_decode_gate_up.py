"""MoE decode gate_up projection alone (Qwen3-30B-A3B shapes: 128 experts, top-8, H 2048, Ip 768), cold weights.
TDIST_MOE_GATE_UP_DOT=0 (decode GEMM on MFMAs) / 1 (vector ALUs)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd import _lib
from triton_distributed_gh_amd.utils.perf import graph_bench

E, H, Ip, topk = 128, 2048, 768, 8
dev, bf = "cuda", torch.bfloat16
lib = _lib.hip()
FORM = int(os.environ.get("TDIST_MOE_GATE_UP_DOT", "-1"))
gu = [((torch.rand(E, 2 * Ip, H, device=dev) * 2 - 1) * 0.05).to(bf) for _ in range(2)]      # 2 x 805 MB
for n in (1, 2, 4, 8, 16):
    x = (torch.rand(n, H, device=dev) * 2 - 1).to(bf)
    hbuf = torch.empty(n * topk, Ip, device=dev, dtype=bf)
    sets = [torch.stack([torch.randperm(E, device=dev)[:topk] for _ in range(n)]).int() for _ in range(40)]
    it = [0]

    def call():
        i = it[0]
        it[0] += 1
        _lib.check(lib.tdist_moe_decode_gate_up(_lib.ptr(x), _lib.ptr(sets[i % 40]), _lib.ptr(gu[i & 1]), _lib.ptr(hbuf), n, topk,
                                                Ip, H, E, FORM, _lib.stream_handle()), "gate_up")

    t = graph_bench(call, n=40)
    mb = n * topk * 2 * Ip * H * 2 / 1e6
    print(f"tokens {n}: {t:.2f} us for {mb:.0f} MB ({mb / t / 1e6:.2f} TB/s)  form={FORM}", flush=True)
