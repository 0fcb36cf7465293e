"""Tile family (256x256, 256x128, 128x256, 128x128) x operand type: element-wise check, then timings.

Usage: probe_tile_family.py [check] [time] [few] [fp8]
  check  every tile shape x dtype on ragged shapes against an fp32 / int64 reference
  time   per-K-step cost of each tile shape (chip full, one and two rounds) - the numbers behind pick_tile()
  few    the few-tiles table (profiles/gemm_few_tiles_r2.md shapes): every tile shape, the cost model's pick, the library
  fp8    4096 x 8192 x 8192 in bf16 / fp8 / int8
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from triton_distributed_gh_amd.ops import gemm as G  # noqa: E402

SHAPES = [(256, 256), (256, 128), (128, 256), (128, 128)]


def bench(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def bench_interleaved(fns, rounds=7, iters=10):
    """median us per call of each fn, A B C A B C ..."""
    ts = [[] for _ in fns]
    for f in fns:
        for _ in range(3):
            f()
    for _ in range(rounds):
        for i, f in enumerate(fns):
            ts[i].append(bench(f, iters=iters, warm=1))
    return [sorted(t)[len(t) // 2] for t in ts]


def make(M, N, K, dt):
    if dt == torch.int8:
        a = torch.randint(-127, 128, (M, K), device="cuda", dtype=torch.int8)
        b = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8)
    else:
        a = ((torch.rand(M, K, device="cuda") * 2 - 1)).to(dt)
        b = ((torch.rand(N, K, device="cuda") * 2 - 1)).to(dt)
    return a, b


def check():
    bad = 0
    for dt in (torch.bfloat16, torch.float16, torch.float8_e4m3fn, torch.float8_e5m2, torch.int8):
        for (M, N, K) in [(512, 512, 512), (300, 264, 384), (1000, 1000, 1024), (130, 136, 256), (257, 520, 640)]:
            a, b = make(M, N, K, dt)
            ref = (a.double() @ b.double().t())
            for bm, bn in SHAPES:
                if dt == torch.int8:
                    o = G.gemm_nt_tiles(a, b, out_dtype=torch.int32, bm=bm, bn=bn)
                    err = (o.double() - ref).abs().max().item()
                    ok = err == 0
                    sa = torch.rand(M, device="cuda") + 0.5
                    sb = torch.rand(N, device="cuda") + 0.5
                    o2 = G.gemm_nt_tiles(a, b, scale_a=sa / 1000, scale_b=sb / 100, bm=bm, bn=bn)
                    ref2 = ref * (sa.double() / 1000)[:, None] * (sb.double() / 100)[None, :]
                    e2 = ((o2.double() - ref2).abs() / (ref2.abs() + 1.0)).max().item()
                    ok = ok and e2 < 1e-2
                    err = max(err, e2)
                else:
                    o = G.gemm_nt_tiles(a, b, bm=bm, bn=bn)
                    tol = 2.0 ** -7 if o.dtype == torch.bfloat16 else 2.0 ** -10
                    e = (o.double() - ref).abs()
                    lim = tol * ref.abs() + 1e-3 * (K ** 0.5)
                    err = (e - lim).max().item()
                    ok = bool((e <= lim).all())
                    sa = torch.rand(M, device="cuda") + 0.5
                    o2 = G.gemm_nt_tiles(a, b, scale_a=sa, bm=bm, bn=bn)
                    ref2 = ref * sa.double()[:, None]
                    e2 = (o2.double() - ref2).abs()
                    ok = ok and bool((e2 <= tol * ref2.abs() + 2e-3 * (K ** 0.5)).all())
                if not ok:
                    bad += 1
                print(f"{str(dt):22s} {M:5d} {N:5d} {K:5d} tile {bm}x{bn}: {'ok' if ok else 'BAD'} ({err:.3g})", flush=True)
    # split-K: every shape, 2..4 ranges, ragged edges, all types (K deep enough for the split)
    for dt in (torch.bfloat16, torch.float8_e4m3fn, torch.int8):
        for (M, N, K) in [(300, 520, 4096), (512, 256, 2048)]:
            a, b = make(M, N, K, dt)
            ref = (a.double() @ b.double().t())
            for bm, bn in SHAPES:
                for sk in (2, 3, 4):
                    for rep in range(2):        # twice: the second launch depends on the counters being re-armed
                        if dt == torch.int8:
                            o = G.gemm_nt_tiles(a, b, out_dtype=torch.int32, bm=bm, bn=bn, split_k=sk)
                            ok = bool((o.double() == ref).all())
                        else:
                            o = G.gemm_nt_tiles(a, b, bm=bm, bn=bn, split_k=sk)
                            e = (o.double() - ref).abs()
                            ok = bool((e <= 2.0 ** -7 * ref.abs() + 1e-3 * (K ** 0.5)).all())
                        if not ok:
                            bad += 1
                        print(f"{str(dt):22s} {M:5d} {N:5d} {K:5d} tile {bm}x{bn} split {sk} #{rep}: {'ok' if ok else 'BAD'}", flush=True)
    print("check:", "all ok" if bad == 0 else f"{bad} BAD")
    return bad


def time_steps():
    dt = torch.bfloat16
    print("per-K-step cost (us/step/workgroup), bf16, N=K-major grid exactly filling 256 CUs x rounds")
    for bm, bn in SHAPES:
        per_cu = 2 if (bm, bn) == (128, 128) else 1
        for rounds in (1, 2):
            for K in (2048, 8192):
                tiles = 256 * per_cu * rounds
                M, N = bm * 16, bn * (tiles // 16)
                a, b = make(M, N, K, dt)
                o = torch.empty(M, N, device="cuda", dtype=dt)
                t = bench(lambda: G.gemm_nt_tiles(a, b, o, bm=bm, bn=bn))
                tf = 2.0 * M * N * K / t / 1e6
                print(f"tile {bm}x{bn} rounds {rounds} K {K}: M {M} N {N}: {t:7.1f} us, {tf:7.0f} TF/s, "
                      f"{t / rounds / (K // 64) * 1000:6.0f} ns/step", flush=True)


def few():
    dt = torch.bfloat16
    rows = []
    variants = [(256, 256, 1), (256, 128, 1), (128, 256, 1), (128, 128, 1), (256, 256, 2), (256, 128, 2), (128, 256, 2),
                (128, 128, 2), (256, 128, 4), (128, 256, 4), (128, 256, 3)]
    for M, N, K in [(2048, 2048, 8192), (2048, 4096, 4096), (2048, 1024, 8192), (1536, 4096, 4096), (4096, 1024, 8192),
                    (2048, 6144, 4096), (3072, 2048, 2048), (4096, 2048, 8192), (4096, 8192, 1024), (4096, 8192, 2048)]:
        a, b = make(M, N, K, dt)
        o = torch.empty(M, N, device="cuda", dtype=dt)
        fns, names = [], []
        for bm, bn, sk in variants:
            tiles = -(-M // bm) * -(-N // bn)
            if sk > 1 and tiles * sk > 256 * (2 if (bm, bn) == (128, 128) else 1):
                continue
            fns.append(lambda bm=bm, bn=bn, sk=sk: G.gemm_nt_tiles(a, b, o, bm=bm, bn=bn, split_k=sk))
            names.append(f"{bm}x{bn}/{sk}")
        fns.append(lambda: G.gemm_nt_tiles(a, b, o))
        fns.append(lambda: torch.matmul(a, b.t(), out=o))
        ts = bench_interleaved(fns)
        rows.append((M, N, K, ts))
        best = min(range(len(names)), key=lambda i: ts[i])
        print(f"{M:5d} {N:5d} {K:5d} | " + " ".join(f"{n} {t:.1f}" for n, t in zip(names, ts)) +
              f" | best {names[best]} {ts[best]:.1f} | model pick {ts[-2]:6.1f} | hipBLASLt {ts[-1]:6.1f}", flush=True)
    wins = sum(1 for r in rows[:7] if r[3][-2] <= r[3][-1])
    print(f"cost-model pick <= library on {wins} of 7 table rows")


def fp8():
    M, N, K = 4096, 8192, 8192
    res = {}
    fns, names = [], []
    keep = []
    for dt in (torch.bfloat16, torch.float8_e4m3fn, torch.float8_e5m2, torch.int8):
        a, b = make(M, N, K, dt)
        o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        keep.append((a, b, o))
        fns.append(lambda a=a, b=b, o=o: G.gemm_nt_tiles(a, b, o, bm=256, bn=256))
        names.append(str(dt))
    a, b, o = keep[0]
    fns.append(lambda: torch.matmul(a, b.t(), out=o))
    names.append("hipBLASLt bf16")
    fns.append(lambda: G.gemm_nt(a, b, o, force_native=True))
    names.append("gemm_nt bf16 (256x256 persistent)")
    ts = bench_interleaved(fns, rounds=9, iters=10)
    for n, t in zip(names, ts):
        print(f"{n:36s} {t:7.1f} us  {2.0 * M * N * K / t / 1e6:7.0f} T(FL)OP/s", flush=True)


def tail():
    """More tiles than CUs, not a whole number of rounds: plan (whole rounds + split last round) vs 256x256 unsplit vs library."""
    dt = torch.bfloat16
    for M, N, K in [(4096, 5120, 25600), (8192, 4608, 36864), (8192, 3584, 14336), (8192, 11008, 4096), (4096, 6400, 5120),
                    (8192, 4096, 14336), (2560, 8192, 8192)]:
        a, b = make(M, N, K, dt)
        o = torch.empty(M, N, device="cuda", dtype=dt)
        ref = None
        fns = [lambda: G.gemm_nt_tiles(a, b, o), lambda: G.gemm_nt_tiles(a, b, o, bm=256, bn=256, split_k=1),
               lambda: torch.matmul(a, b.t(), out=o)]
        G.gemm_nt_tiles(a, b, o)
        ref = a[:256].double() @ b.double().t()
        ok = bool(((o[:256].double() - ref).abs() <= 2.0 ** -7 * ref.abs() + 1e-3 * K ** 0.5).all())
        ts = bench_interleaved(fns, rounds=5, iters=5)
        print(f"{M:5d} {N:5d} {K:5d} tiles(256) {-(-M // 256) * -(-N // 256):4d} | plan {ts[0]:7.1f} | 256x256/1 {ts[1]:7.1f} | "
              f"hipBLASLt {ts[2]:7.1f} | first rows {'ok' if ok else 'BAD'}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time", "few", "fp8"]
    if "tail" in what:
        tail()
    rc = 0
    if "check" in what:
        rc = check()
    if "time" in what:
        time_steps()
    if "few" in what:
        few()
    if "fp8" in what:
        fp8()
    sys.exit(1 if rc else 0)
