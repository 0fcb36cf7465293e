"""Builds the table of measured GEMM forms (triton_distributed_gh_amd/tuned/gemm_mi355x.json, read by ops/gemm.py).

For every weight shape (N, K) of the chosen model families and one row count per bucket, every kernel the package has
for that row count is timed on COLD weights (copies of the matrix rotated inside a HIP graph, as a model step sees
them) next to what gemm_nt's rules pick; a form enters the table where it beats the rules by more than MARGIN.
    python scripts/tune_gemm_table.py [--families a,b] [--shapes "N,K;N,K"] [--out file.json] [--margin 0.04] [--merge]
"""
import argparse, ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triton_distributed_gh_amd.ops import gemm as G
from triton_distributed_gh_amd.utils.perf import graph_bench

FAMILIES = {          # (N, K) of qkv / o / gate_up / down (+ LM head)
    "qwen3-8b": [(6144, 4096), (4096, 4096), (24576, 4096), (4096, 12288), (151936, 4096)],
    "qwen3-32b": [(10240, 5120), (5120, 8192), (51200, 5120), (5120, 25600), (151936, 5120)],
    "llama3-8b": [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (128256, 4096)],
    "llama3-70b": [(10240, 8192), (8192, 8192), (57344, 8192), (8192, 28672), (128256, 8192)],
    "llama3-70b-tp8": [(1280, 8192), (8192, 1024), (7168, 8192), (8192, 3584)],
    "qwen3-32b-tp8": [(1280, 5120), (5120, 1024), (6400, 5120), (5120, 3200)],
    "qwen3-14b": [(7168, 5120), (5120, 5120), (34816, 5120), (5120, 17408)],
    "seed-oss-36b": [(12288, 5120), (5120, 10240), (55296, 5120), (5120, 27648), (155136, 5120)],
    "seed-oss-36b-tp8": [(1536, 5120), (5120, 1280), (6912, 5120), (5120, 3456)],
    "qwen3-30b-a3b-attn": [(5120, 2048), (2048, 4096), (151936, 2048)],
    "qwen2.5-7b": [(4608, 3584), (3584, 3584), (37888, 3584), (3584, 18944), (152064, 3584)],
    "qwen2.5-72b": [(59136, 8192), (8192, 29568), (152064, 8192)],
    "mixtral-8x7b-attn": [(6144, 4096), (4096, 4096), (32000, 4096)],
    "llama3.2-3b": [(5120, 3072), (3072, 3072), (16384, 3072), (3072, 8192), (128256, 3072)],
    "gemma2-9b": [(8192, 3584), (3584, 4096), (28672, 3584), (3584, 14336), (256000, 3584)],
    "mistral-nemo-12b": [(6144, 5120), (5120, 4096), (28672, 5120), (5120, 14336), (131072, 5120)],
    "llama3.1-405b-tp8": [(2304, 16384), (16384, 2048), (13312, 16384), (16384, 6656)],
    "qwen3-235b-a22b-attn": [(9216, 4096), (4096, 8192)],
}


def candidates(M, N, K):
    c = []
    if M == 1:
        c += [f"dot:{nw}" for nw in (1, 2, 4, 8)]
    if 2 <= M <= 16:
        c += [f"s128:{wn}{nw}" for wn, nw in ((1, 4), (2, 4), (4, 4), (2, 8), (4, 8), (8, 8))]
    if 2 <= M <= 1024:
        c += ["mid"]
    if 17 <= M <= 256:
        c += ["mid:1:0", "mid:2:0"]
    if M >= 17:
        c += ["tiles", "tile:128x128", "tile:128x256", "tile:256x128", "tile:256x256", "native256"]
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", default=",".join(FAMILIES))
    ap.add_argument("--out", default=os.path.join(os.path.dirname(G.__file__), "..", "tuned", "gemm_mi355x.json"))
    ap.add_argument("--margin", type=float, default=0.04)
    ap.add_argument("--merge", action="store_true", help="keep the cells of an existing output file for shapes not tuned now")
    ap.add_argument("--rows", default=",".join(str(v) for v in G.GEMM_BUCKETS))
    ap.add_argument("--shapes", default="", help='further weight shapes "N,K;N,K" (with --families "" only these)')
    args = ap.parse_args()
    dev, bf = "cuda", torch.bfloat16
    shapes = []
    for fam in [f for f in args.families.split(",") if f]:
        for s in FAMILIES[fam]:
            if s not in shapes:
                shapes.append(s)
    for s in [tuple(int(v) for v in t.split(",")) for t in args.shapes.split(";") if t]:
        if s not in shapes:
            shapes.append(s)
    table, log = {}, []
    for N, K in shapes:
        copies = max(2, min(8, int(700e6 / (N * K * 2)) + 1))
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(copies)]
        for M in [int(v) for v in args.rows.split(",")]:
            x = torch.randn(M, K, device=dev).to(bf)
            out = torch.empty(M, N, device=dev, dtype=bf)
            ref = x.float() @ ws[0].float().t()
            scale = ref.abs().max().item()

            def timed(fn):
                it = [0]
                def f():
                    i = it[0]; it[0] += 1
                    fn(ws[i % copies])
                f()
                return graph_bench(f, n=copies * 2)

            rules = lambda w: G.gemm_nt(x, w, out=out, use_table=False)
            timed(rules)                                  # (settles clocks after the fp32 reference product)
            t_rules = timed(rules)
            best, t_best = None, t_rules
            for form in candidates(M, N, K):
                try:
                    if not G.run_gemm_form(form, x, ws[0], out):
                        continue
                    if not (out.float() - ref).abs().max().item() <= 2e-2 * scale:
                        log.append(f"{N}x{K} M={M} {form}: WRONG RESULT")
                        continue
                    t = timed(lambda w: G.run_gemm_form(form, x, w, out))
                except Exception as exc:  # noqa: BLE001
                    log.append(f"{N}x{K} M={M} {form}: {type(exc).__name__}: {str(exc)[:60]}")
                    continue
                if t < t_best:
                    best, t_best = form, t
            t_rules = min(t_rules, timed(rules))
            line = f"{N}x{K} M={M}: rules {t_rules:.1f} us, best {best or 'rules'} {t_best:.1f} us"
            if best and t_best < (1 - args.margin) * t_rules:
                table.setdefault(f"{N},{K}", {})[str(M)] = best
                line += "  -> table"
            print(line, flush=True)
        del ws
    if args.merge and os.path.exists(args.out):
        with open(args.out) as f:
            old = json.load(f)
        for k, v in old.items():
            if "," in k and k not in [f"{n},{kk}" for n, kk in shapes]:
                table[k] = v
    with open(args.out, "w") as f:
        json.dump({"_about": "measured GEMM forms per (N,K) and row bucket on MI355X, bf16, cold weights; written by "
                             "scripts/tune_gemm_table.py", **table}, f, indent=1, sort_keys=True)
    print(f"{sum(len(v) for v in table.values())} cells written to {args.out}")
    for l in log:
        print("note:", l)


if __name__ == "__main__":
    main()
