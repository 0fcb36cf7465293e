"""Hammers the wait-free split-K join of the fused AG + GEMM kernel on one rank (no peers): ITER launches per shape
and split, every result compared with the fp32 product, optionally with a copy loop on a second stream that keeps
the memory system busy (LOAD=1).  Prints the number of launches whose result was off and the worst relative error.
TDIST_SAFE_PUBLISH=1 runs the release-fence flavour of the library."""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ["TDIST_FORCE_FUSED"] = "1"
iters = int(os.environ.get("ITER", "2000"))
load = os.environ.get("LOAD") == "1"
from triton_distributed_gh_amd.ops.ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context  # noqa: E402
from triton_distributed_gh_amd.utils.testing import VirtualCluster  # noqa: E402

side = torch.cuda.Stream()
big_a = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
big_b = torch.empty_like(big_a)
for (M, N, K) in ((520, 776, 4096), (1024, 1024, 8192), (4096, 1024, 8192)):
    for split in (2, 3, 4):
        os.environ["TDIST_AG_SPLIT_K"] = str(split)
        vc = VirtualCluster(1, heap_bytes=1 << 28)
        try:
            ctx = create_ag_gemm_intra_node_context(M, N, K, torch.bfloat16, torch.bfloat16, 0, 1, env=vc.envs[0])
            bad, worst, first = 0, 0.0, -1
            for it in range(iters):
                if it % 50 == 0:
                    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
                    b = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
                    ref = a.float() @ b.float().t()
                    scale = ref.abs().max().item()
                    torch.cuda.synchronize()
                if load and it % 4 == 0:
                    with torch.cuda.stream(side):
                        big_b.copy_(big_a, non_blocking=True)
                out = vc.run(lambda r, env: ag_gemm_intra_node(a, b, False, ctx))[0]
                rel = (out.float() - ref).abs().max().item() / scale
                if not rel < 1e-2:
                    bad += 1
                    first = it if first < 0 else first
                worst = max(worst, rel if rel == rel else float("inf"))
            ctx.check_error()
            torch.cuda.synchronize()
            print(f"M={M} N={N} K={K} split={split} load={int(load)}: {bad} of {iters} launches off (first {first}), worst rel {worst:.2e}", flush=True)
        finally:
            vc.close()
