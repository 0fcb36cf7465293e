"""Race hunt: AG-GEMM and GEMM-RS, hundreds of fresh-input iterations in the hostile-timing debug mode.

    TDIST_DEBUG_RACES=1 python triton_dist/test/stress/stress_race_debug.py --virtual-ranks 4 --iters 200

With TDIST_DEBUG_RACES=1 every producer sleeps a random time before it stores a payload and again before it
raises the flag, every kernel checks its parity's flags on entry and the host poisons a parity buffer with NaNs
once its call has consumed it (csrc/include/tdist/race_debug.hpp).  Each iteration draws new inputs (and, from
a short list, a new row count - whole tiles, partial tiles and rows that straddle ranks), one random rank starts
late, and every output element is checked against an fp32 reference: atol 1e-3 + 2e-2 of the magnitude that
was rounded to bf16 on the way (one bf16 ulp, 2^-7 relative at worst, per rounded partial plus the final rounding).

Scenario of the reference's repeat-and-perturb tests (reference:
python/triton_dist/test/amd/test_distributed-notify-wait.py:39-212 - 2000 repeats,
test/nvidia/test_allreduce.py:185-197 --verify_hang / --stress / --simulate_straggler,
kernels/nvidia/allgather.py:74-78 random producer noise).
"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.kernels.amd import (ag_gemm_intra_node, create_ag_gemm_intra_node_context,  # noqa: E402
                                     create_gemm_rs_intra_node_context, gemm_rs_intra_node)
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args  # noqa: E402


def off_elements(got, gold, mag):
    return int(((got.float() - gold).abs() > 1e-3 + 2e-2 * mag).sum()) + int((~torch.isfinite(got.float())).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=1024)
    ap.add_argument("--K", type=int, default=512)
    add_common_args(ap)
    ap.set_defaults(iters=200)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, dt = R.world_size, R.device, torch.bfloat16
    assert R.virtual, "run with --virtual-ranks N (the per-iteration reference needs every rank's inputs)"
    N, K = args.N, args.K
    row_choices = [256 * w, 512 * w, 64 * w, 16 * w, 320 * w]       # rows per rank: 256, 512, 64, 16, 320
    max_M = max(row_choices)
    ctas = max(8, 200 // w)
    ag = R.collective(lambda r, env: create_ag_gemm_intra_node_context(max_M, N // w, K, dt, dt, r, w, num_ctas=ctas, env=env))
    rs = R.collective(lambda r, env: create_gemm_rs_intra_node_context(max_M, N, dt, r, w, num_ctas=ctas, env=env))
    print(f"race debug: {'on' if os.environ.get('TDIST_DEBUG_RACES') == '1' else 'off'}; {w} ranks, {args.iters} iterations",
          flush=True)
    rng = random.Random(args.seed)
    for it in range(args.iters):
        M = rng.choice(row_choices)
        Mr = M // w
        late = rng.randrange(w)
        a = [(torch.rand(Mr, K, device=dev) * 2 - 1).to(dt) for _ in range(w)]
        b = [(torch.rand(N // w, K, device=dev) * 2 - 1).to(dt) for _ in range(w)]
        a2 = [(torch.rand(M, K // w, device=dev) * 2 - 1).to(dt) for _ in range(w)]
        b2 = [(torch.rand(N, K // w, device=dev) * 2 - 1).to(dt) for _ in range(w)]
        torch.cuda.synchronize()

        def step(r, env):
            if r == late:
                torch.cuda._sleep(rng.randrange(1, 4) * 2_000_000)
            c1 = ag_gemm_intra_node(a[r], b[r], False, ag[r])
            c2 = gemm_rs_intra_node(a2[r], b2[r], rs[r])
            return c1, c2

        outs = R.run(step)
        a_full = torch.cat(a).float()
        parts = [(a2[r].float() @ b2[r].float().t()).to(dt).float() for r in range(w)]
        gold2, mag2 = sum(parts), sum(p.abs() for p in parts)
        for r in range(w):
            ag[r].check_error()
            rs[r].check_error()
            g1 = a_full @ b[r].float().t()
            n1 = off_elements(outs[r][0], g1, g1.abs())
            n2 = off_elements(outs[r][1], gold2[r * Mr:(r + 1) * Mr], mag2[r * Mr:(r + 1) * Mr])
            assert n1 == 0 and n2 == 0, f"iteration {it} (M={M}, late rank {late}) rank {r}: {n1} AG-GEMM / {n2} GEMM-RS elements off"
        if (it + 1) % 50 == 0:
            print(f"  {it + 1} iterations ok", flush=True)
    print(f"✅ race hunt passed: {args.iters} iterations x (AG-GEMM + GEMM-RS) on {w} ranks", flush=True)
    R.close()


if __name__ == "__main__":
    main()
