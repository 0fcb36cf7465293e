"""Decode attention (one page per sequence, 32 q heads / 8 kv heads, D = 128) against the number of KV splits."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.ops import flash_decode

def bench(fn, n=20, reps=10):
    """n calls captured in one HIP graph: kernel time without the host's launch cost."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3

dev, bf, Hq, Hkv, D = "cuda", torch.bfloat16, 32, 8, 128
ROWS = [(1, 512), (1, 2048), (1, 4096), (1, 8192), (1, 32768), (1, 131072), (2, 2048), (2, 32768), (4, 2048), (4, 8192), (4, 32768), (8, 8192), (16, 4096)]
if os.environ.get("ROWS"):                 # e.g. ROWS=1x8192,1x32768
    ROWS = [tuple(int(v) for v in r.split("x")) for r in os.environ["ROWS"].split(",")]
for B, L in ROWS:
    per = 2 * B * L * Hkv * D * 2
    copies = max(2, min(160, int(600e6 / per) + 1))          # cold K / V on every call, like a layer of a model
    kcs = [torch.randn(B, L, Hkv, D, device=dev).to(bf) for _ in range(copies)]
    vcs = [torch.randn(B, L, Hkv, D, device=dev).to(bf) for _ in range(copies)]
    q = torch.randn(B, Hq, D, device=dev).to(bf)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    table = torch.arange(B, dtype=torch.int32, device=dev).view(B, 1)
    auto = flash_decode.auto_kv_splits(B, Hq, Hkv, L, fused=True)
    base = B * Hkv
    cands = sorted({1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64, auto, max(1, 256 // base), max(1, 512 // base)})
    cands = [c for c in cands if c <= max(1, L // 64) and c <= 64]
    res = []
    for sp in cands:
        it = [0]

        def call():
            i = it[0] % copies
            it[0] += 1
            flash_decode.gqa_fwd_batch_decode_persistent(q, kcs[i], vcs[i], None, None, lens, table, D ** -0.5, kv_split=sp)

        it[0] = 0
        t = bench(call, n=min(copies, 40))
        res.append(f"{sp}{'*' if sp == auto else ''}:{t:.1f}")
    # the two-launch form (split kernel, then a merge kernel that is parallel over heads)
    two = []
    for sp in sorted({flash_decode.auto_kv_splits(B, Hq, Hkv, L), max(1, 256 // base), max(1, 512 // base), 16, 32, 64}):
        if sp > max(1, L // 64) or sp > 64:
            continue
        it = [0]

        def call2():
            i = it[0] % copies
            it[0] += 1
            flash_decode.gqa_fwd_batch_decode(q, kcs[i], vcs[i], None, None, lens, table, D ** -0.5, kv_split=sp)

        two.append(f"{sp}:{bench(call2, n=min(copies, 40)):.1f}")
    print(f"B={B} L={L} base={base} copies={copies}: " + "  ".join(res) + "   | two launches: " + "  ".join(two), flush=True)
    del kcs, vcs

# capacity far above the attended length: the split count of the grid against the count the sequence uses
print("capacity 8192, one sequence: attended length -> us (automatic / same grid, every split used)")
B, L = 1, 8192
kcs = [torch.randn(B, L, Hkv, D, device=dev).to(bf) for _ in range(18)]
vcs = [torch.randn(B, L, Hkv, D, device=dev).to(bf) for _ in range(18)]
q = torch.randn(B, Hq, D, device=dev).to(bf)
table = torch.arange(B, dtype=torch.int32, device=dev).view(B, 1)
auto = flash_decode.auto_kv_splits(B, Hq, Hkv, L, fused=True)
for n in (100, 300, 1000, 3000, 8192):
    lens = torch.full((B,), n, dtype=torch.int32, device=dev)
    out = []
    for sp in (-1, auto):
        it = [0]

        def call():
            i = it[0] % 18
            it[0] += 1
            flash_decode.gqa_fwd_batch_decode_persistent(q, kcs[i], vcs[i], None, None, lens, table, D ** -0.5, kv_split=sp)

        out.append(bench(call, n=18))
    print(f"  len {n}: {out[0]:.1f} / {out[1]:.1f}", flush=True)
