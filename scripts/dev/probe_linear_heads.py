"""The LM-head shapes of my_lab/benchmark/bench_matmul.sh through gemm_nt against torch.matmul, without the bias add:
warm (one weight matrix called back to back, as that benchmark does - part of it stays in the 256 MB memory-side cache)
and cold (weight copies rotated so that every call streams from HBM, as a decode step does).  us per call, HIP graph
replays.  Kernel choices come from the environment (TDIST_SKINNY128, TDIST_DECODE_DOT, TDIST_DECODE_SPLITK_MIN_M ...)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.ops.gemm import gemm_nt
from triton_distributed_gh_amd.utils.perf import graph_bench
dev, bf = "cuda", torch.bfloat16
shapes = [("mistral-7b", 4096, 32768), ("llama3-8b", 4096, 128256), ("llama3-70b", 8192, 12800), ("qwen3-30b", 2048, 151936),
          ("gpt-neo", 6144, 50251), ("Gemma-7b", 3072, 256000)]
only = os.environ.get("SHAPES")
rows = [int(v) for v in os.environ.get("ROWS", "1,8,16,32,64,128").split(",")]
tag = " ".join(f"{k[6:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("TDIST_"))
for name, K, N in shapes:
    if only and name not in only.split(","):
        continue
    nbytes = N * K * 2
    copies = max(2, min(8, int(700e6 / nbytes) + 1))
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(copies)]
    for M in rows:
        x = torch.randn(M, K, device=dev).to(bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        res = {}
        for mode in ("warm", "cold"):
            for impl in ("native", "torch"):
                it = [0]
                def f():
                    i = it[0]; it[0] += 1
                    w = ws[i % copies] if mode == "cold" else ws[0]
                    if impl == "native":
                        gemm_nt(x, w, out=out)
                    else:
                        torch.matmul(x, w.t(), out=out)
                f()
                res[mode, impl] = graph_bench(f, n=copies * 2)
        gemm_nt(x, ws[0], out=out)
        ref = x.float() @ ws[0].float().t()
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        print(f"{name} {K}->{N} M={M} [{tag}]: warm native {res['warm','native']:.1f} torch {res['warm','torch']:.1f} | "
              f"cold native {res['cold','native']:.1f} torch {res['cold','torch']:.1f}" + ("" if err < 2e-2 else f" (ERR {err:.1e})"),
              flush=True)
    del ws
