"""Mid-M GEMM (csrc/ops/gemm_mid.hip) against hipBLASLt (torch.matmul) over model-shaped layers.

    python scripts/dev/probe_mid_gemm.py            # table: M x (N, K), native us, library us, ratio
    python scripts/dev/probe_mid_gemm.py scan       # plan scan (wn, splits) for a few shapes
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from triton_distributed_gh_amd.ops.gemm import gemm_nt_mid  # noqa: E402

SHAPES = [  # (name, N, K)
    ("qwen3-32b o_proj/TP8", 5120, 1024), ("qwen3-32b down/TP8", 5120, 3200), ("qwen3-32b gate_up/TP8", 6400, 5120),
    ("qwen3-8b qkv", 6144, 4096), ("qwen3-8b o_proj", 4096, 4096), ("qwen3-8b gate_up", 24576, 4096),
    ("qwen3-8b down", 4096, 12288), ("llama3-70b gate_up/TP8", 7168, 8192), ("llama3-70b down/TP8", 8192, 3584),
    ("lm_head/TP8", 18992, 4096),
]


def timeit(fn, n=40, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dt = torch.bfloat16
    if len(sys.argv) > 1 and sys.argv[1] == "scan":
        for M in (32, 64, 128):
            for name, N, K in [SHAPES[i] for i in (2, 5, 6, 7, 9)]:
                x = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt)
                w = (torch.rand(N, K, device="cuda") * 2 - 1).to(dt)
                o = torch.empty(M, N, device="cuda", dtype=dt)
                row = []
                for wn in (1, 2):
                    for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                        if sp > K // 128:
                            continue
                        row.append(f"wn{wn}/s{sp}:{timeit(lambda: gemm_nt_mid(x, w, o, wn=wn, splits=sp)):6.1f}")
                lib = timeit(lambda: torch.matmul(x, w.t(), out=o))
                print(f"M={M:4d} {name:26s} lib {lib:6.1f} | " + " ".join(row), flush=True)
        return
    wins = total = 0
    print(f"{'layer':28s} {'N':>6s} {'K':>6s} " + " ".join(f"{'M=' + str(m):>22s}" for m in (32, 64, 128)))
    for name, N, K in SHAPES:
        cells = []
        for M in (32, 64, 128):
            x = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt)
            w = (torch.rand(N, K, device="cuda") * 2 - 1).to(dt)
            o = torch.empty(M, N, device="cuda", dtype=dt)
            # interleave so both see the same clocks / cache state
            a = min(timeit(lambda: gemm_nt_mid(x, w, o)) for _ in range(2))
            b = min(timeit(lambda: torch.matmul(x, w.t(), out=o)) for _ in range(2))
            wins += a <= b
            total += 1
            cells.append(f"{a:6.1f} / {b:6.1f} ({b / a:4.2f}x)")
        print(f"{name:28s} {N:6d} {K:6d} " + " ".join(f"{c:>22s}" for c in cells), flush=True)
    print(f"native <= library on {wins} of {total} cells")


if __name__ == "__main__":
    main()
