"""Every kernel this package has for a linear layer with few rows, on LM-head and layer shapes: which one is fastest
per (shape, rows)?  Cold (weight copies rotated, every call streams from HBM) and warm (one matrix back to back).
us per call from HIP-graph replays.  SHAPES=name,name and ROWS=1,8,... narrow the sweep."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd import _lib
from triton_distributed_gh_amd.ops.gemm import gemm_nt, gemm_nt_mid, gemm_nt_tiles
from triton_distributed_gh_amd.utils.perf import graph_bench
dev, bf = "cuda", torch.bfloat16
lib = _lib.hip()
shapes = [("mistral-7b", 4096, 32768), ("llama3-8b", 4096, 128256), ("llama3-70b", 8192, 12800), ("qwen3-30b", 2048, 151936),
          ("Gemma-7b", 3072, 256000), ("q8b-qkv", 4096, 6144), ("q8b-o", 4096, 4096), ("q8b-down", 12288, 4096),
          ("q8b-gate_up", 4096, 24576)]
only = os.environ.get("SHAPES")
rows = [int(v) for v in os.environ.get("ROWS", "1,2,4,8,16,32,64,128").split(",")]
LL = ctypes.c_longlong


def candidates(M, N, K):
    c = {"default": lambda x, w, o: gemm_nt(x, w, out=o), "torch": lambda x, w, o: torch.matmul(x, w.t(), out=o)}
    if M <= 2:
        for nw in (1, 2, 4, 8):
            c[f"dot{nw}"] = (lambda nw: lambda x, w, o: _lib.check(lib.tdist_gemm_nt_dot(
                _lib.ptr(x), _lib.ptr(w), _lib.ptr(o), M, N, K, LL(K), LL(K), LL(N), 0, nw, _lib.stream_handle()), "dot"))(nw)
    if 2 <= M <= 16:
        for wn, nw in ((1, 4), (2, 4), (4, 4), (2, 8), (4, 8), (8, 8)):
            c[f"s128-{wn}{nw}"] = (lambda wn, nw: lambda x, w, o: _lib.check(lib.tdist_gemm_nt_skinny128(
                _lib.ptr(x), _lib.ptr(w), _lib.ptr(o), M, N, K, LL(K), LL(K), LL(N), wn, nw, _lib.stream_handle()), "s128"))(wn, nw)
    if M >= 2:
        c["mid"] = lambda x, w, o: gemm_nt_mid(x, w, o)
    if M >= 17 and os.environ.get("MID_FORMS") == "1":
        for wn, sp in ((1, 1), (1, 2), (2, 2), (2, 3), (2, 4), (1, 4)):
            c[f"mid-w{wn}s{sp}"] = (lambda wn, sp: lambda x, w, o: gemm_nt_mid(x, w, o, wn=wn, splits=sp))(wn, sp)
    if M >= 32:
        for bm, bn in ((128, 128), (128, 256), (256, 256)):
            c[f"tile{bm}x{bn}"] = (lambda bm, bn: lambda x, w, o: gemm_nt_tiles(x, w, o, bm=bm, bn=bn))(bm, bn)
    if M >= 160:
        del c["mid"]
        c["tile256x128"] = lambda x, w, o: gemm_nt_tiles(x, w, o, bm=256, bn=128)
        c["tiles-auto"] = lambda x, w, o: gemm_nt_tiles(x, w, o)
        for sk in (2, 3, 4):
            c[f"tile128x256/{sk}"] = (lambda sk: lambda x, w, o: gemm_nt_tiles(x, w, o, bm=128, bn=256, split_k=sk))(sk)
            c[f"tile256x256/{sk}"] = (lambda sk: lambda x, w, o: gemm_nt_tiles(x, w, o, bm=256, bn=256, split_k=sk))(sk)
        c["native256"] = lambda x, w, o: gemm_nt(x, w, out=o, force_native=True)
        if M <= 1024:
            c["mid"] = lambda x, w, o: gemm_nt_mid(x, w, o)
    return c


for name, K, N in shapes:
    if only and name not in only.split(","):
        continue
    copies = max(2, min(16, int(700e6 / (N * K * 2)) + 1))
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(copies)]
    for M in rows:
        x = torch.randn(M, K, device=dev).to(bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        ref = x.float() @ ws[0].float().t()
        line = {}
        for cname, fn in candidates(M, N, K).items():
            try:
                fn(x, ws[0], out)
                err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
                res = []
                for mode in ("cold", "warm"):
                    it = [0]
                    def f():
                        i = it[0]; it[0] += 1
                        fn(x, ws[i % copies] if mode == "cold" else ws[0], out)
                    res.append(graph_bench(f, n=copies * 2))
                line[cname] = (res[0], res[1], err)
            except Exception as exc:  # noqa: BLE001
                line[cname] = (float("inf"), float("inf"), str(exc)[:40])
        best = min((k for k in line if k not in ("default", "torch")), key=lambda k: line[k][0])
        txt = "  ".join(f"{k} {v[0]:.1f}/{v[1]:.1f}" + ("" if not isinstance(v[2], str) and v[2] < 2e-2 else f"(ERR {v[2]})")
                        for k, v in line.items())
        print(f"{name} {N}x{K} M={M}: best cold {best} | {txt}", flush=True)
    del ws
