"""Decode GEMMs of a Qwen3-8B layer + LM head at 1 and 2 rows: the vector-ALU form (tdist_gemm_nt_dot) against what
gemm_nt / gemm_nt_silu_mul pick (TDIST_DECODE_DOT=0: the MFMA decode kernel), inside a HIP graph over rotating weight
copies (cold weights)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd import _lib
from triton_distributed_gh_amd.ops.gemm import gemm_nt
from triton_distributed_gh_amd.utils.perf import graph_bench

dev, bf = "cuda", torch.bfloat16
lib = _lib.hip()
shapes = [("qkv", 6144, 4096, 0), ("gate_up+silu", 12288, 4096, 1), ("o_proj", 4096, 4096, 0), ("down", 4096, 12288, 0),
          ("lm_head", 151936, 4096, 0), ("moe-router", 128, 2048, 0)]
for name, N, K, silu in shapes:
    rows = 2 * N if silu else N
    copies = max(2, min(24, int(700e6 / (rows * K * 2)) + 1))
    ws = [(torch.randn(rows, K, device=dev) * 0.02).to(bf) for _ in range(copies)]
    line = []
    for M in (1, 2):
        x = torch.randn(M, K, device=dev).to(bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        it = [0]

        def dot(nw):
            def f():
                i = it[0]; it[0] += 1
                _lib.check(lib.tdist_gemm_nt_dot(_lib.ptr(x), _lib.ptr(ws[i % copies]), _lib.ptr(out), M, N, K, ctypes.c_longlong(K),
                                                 ctypes.c_longlong(K), ctypes.c_longlong(N), silu, nw, _lib.stream_handle()), "dot")
            return f

        def mfma():
            i = it[0]; it[0] += 1
            if silu:
                _lib.check(lib.tdist_gemm_nt_silu_mul(_lib.ptr(x), _lib.ptr(ws[i % copies]), _lib.ptr(out), M, N, K, ctypes.c_longlong(K),
                                                      ctypes.c_longlong(K), ctypes.c_longlong(N), _lib.stream_handle()), "silu")
            else:
                gemm_nt(x, ws[i % copies], out=out)

        dot(0)()          # correctness against fp32
        w0 = ws[(it[0] - 1) % copies].float()
        ref = x.float() @ w0.t()
        if silu:
            ref = torch.nn.functional.silu(ref[:, :N].to(bf).float()) * ref[:, N:].to(bf).float()
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        t_m = graph_bench(mfma, n=copies)
        ts = {nw: graph_bench(dot(nw), n=copies) for nw in (0, 1, 2, 4, 8)}
        line.append(f"M={M}: lib's choice {t_m:.1f} | dot auto {ts[0]:.1f} nw1 {ts[1]:.1f} nw2 {ts[2]:.1f} nw4 {ts[4]:.1f} nw8 {ts[8]:.1f} (err {err:.1e})")
    print(f"{name} {N}x{K} ({rows * K * 2 / 1e6:.0f} MB): " + "  ||  ".join(line), flush=True)
    del ws
