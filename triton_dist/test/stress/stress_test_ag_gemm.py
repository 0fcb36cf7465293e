"""AG-GEMM stress: random M every iteration, random straggler rank, hang watchdog.

    bash scripts/launch_amd.sh python/triton_dist/test/stress/stress_test_ag_gemm.py --iters 200 --verify_hang 40 --simulate_straggler

(Scenario of reference python/triton_dist/test/stress/stress_test_ag_gemm.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
os.environ.setdefault("RANDOM_PARAMS", "1")

import torch  # noqa: E402

import random  # noqa: E402
import threading  # noqa: E402
import time  # noqa: E402

from triton_dist.kernels.amd import ag_gemm_intra_node, create_ag_gemm_intra_node_context  # noqa: E402
from triton_distributed_gh_amd.utils.cli import Runner, add_common_args, rel_err  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max_M", type=int, default=8192)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--K", type=int, default=2048)
    ap.add_argument("--verify_hang", type=int, default=40, help="seconds without progress before the run aborts")
    ap.add_argument("--simulate_straggler", action="store_true")
    add_common_args(ap)
    ap.set_defaults(iters=100)
    args = ap.parse_args()
    R = Runner(args)
    w, dev, dt = R.world_size, R.device, torch.bfloat16
    ctxs = R.collective(lambda r, env: create_ag_gemm_intra_node_context(args.max_M, args.N // w, args.K, dt, dt, r, w,
                                                                        env.tp_group, env=env))
    progress = [time.time()]

    def watchdog():
        while True:
            time.sleep(1.0)
            if time.time() - progress[0] > args.verify_hang:
                print(f"❌ no progress for {args.verify_hang}s: hang suspected", flush=True)
                os._exit(3)

    threading.Thread(target=watchdog, daemon=True).start()
    rng = random.Random(args.seed)            # same sequence on every rank
    for it in range(args.iters):
        M = rng.randrange(w, args.max_M // w + 1) // 8 * 8 * w or 8 * w
        straggler = rng.randrange(w) if args.simulate_straggler else -1
        A = {r: (torch.rand(M // w, args.K, device=dev) * 2 - 1).to(dt) for r in R.local_ranks}
        B = {r: (torch.rand(args.N // w, args.K, device=dev) * 2 - 1).to(dt) for r in R.local_ranks}
        R.barrier()

        def fn(r, env):
            if r == straggler and env.backend == "hip":
                torch.cuda._sleep(rng.randrange(1, 50) * 1_000_000)
            return ag_gemm_intra_node(A[r], B[r], False, ctxs[r])

        out = R.run(fn)
        full = torch.cat(R.all_gather(A)).float()
        for r in R.local_ranks:
            ctxs[r].check_error()
            assert rel_err(out[r], full @ B[r].float().t()) < 1e-2, f"iteration {it} M={M} rank {r}"
        progress[0] = time.time()
    R.print(f"✅ stress: {args.iters} iterations with random M passed on {w} ranks")
    R.close()


if __name__ == "__main__":
    main()
