"""Decode attention at batch 1 with cold K / V (32 q heads / 8 kv heads, D = 128, two-launch form): us and TB/s
against the number of KV splits.  TDIST_DECODE_DEEP=0/1 in the environment forces the register-set choice."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from triton_distributed_gh_amd.ops import flash_decode
from triton_distributed_gh_amd.utils.perf import graph_bench as bench  # noqa: E402

dev, bf, Hq, Hkv, D = "cuda", torch.bfloat16, 32, 8, 128
page = int(os.environ.get("PAGE", "0"))
for B, L in [(1, 8192), (1, 32768), (1, 131072)]:
    per = 2 * B * L * Hkv * D * 2
    copies = max(3, min(40, int(1200e6 / per) + 1))
    kcs = [torch.randn(B, L, Hkv, D, device=dev).to(bf) for _ in range(copies)]
    vcs = [torch.randn(B, L, Hkv, D, device=dev).to(bf) for _ in range(copies)]
    q = torch.randn(B, Hq, D, device=dev).to(bf)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    if page:
        table = torch.arange(B * L // page, dtype=torch.int32, device=dev).view(B, -1)
        kcs = [k.view(-1, page, Hkv, D) for k in kcs]
        vcs = [v.view(-1, page, Hkv, D) for v in vcs]
    else:
        table = torch.arange(B, dtype=torch.int32, device=dev).view(B, 1)
    out = []
    for sp in (-1, 16, 24, 32, 48, 64, 96):
        if sp > 0 and sp > L // 128:
            continue
        it = [0]

        def call():
            i = it[0] % copies
            it[0] += 1
            flash_decode.gqa_fwd_batch_decode(q, kcs[i], vcs[i], None, None, lens, table, D ** -0.5, kv_split=sp)

        t = bench(call, n=min(copies, 20))
        out.append(f"{sp}:{t:.1f}us/{per / t / 1e6:.2f}TB/s")
    auto = flash_decode.auto_kv_splits(B, Hq, Hkv, L)
    print(f"B={B} L={L} page={page or L} deep={os.environ.get('TDIST_DECODE_DEEP', 'auto')} auto={auto}: " + "  ".join(out), flush=True)
    del kcs, vcs
