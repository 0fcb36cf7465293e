"""Dense GEMMs with the FLOPs and shapes of the two Mixtral grouped GEMMs (one rank of 8): what the tile engine and
the library reach without gather, flags and padding - the bound for the grouped kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from triton_distributed_gh_amd.ops import gemm as G
from probe_tile_family import bench_interleaved  # noqa: E402

for name, (M, N, K) in (("AG+MoE  [32768 x 4096] x [1792 x 4096]^T", (32768, 1792, 4096)),
                        ("MoE+RS  [32768 x 1792] x [4096 x 1792]^T", (32768, 4096, 1792))):
    a = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
    b = (torch.rand(N, K, device="cuda") - 0.5).bfloat16()
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ts = bench_interleaved([lambda: G.gemm_nt(a, b, o, force_native=True), lambda: G.gemm_nt_tiles(a, b, o),
                            lambda: torch.matmul(a, b.t(), out=o)], rounds=5, iters=5)
    fl = 2.0 * M * N * K
    print(f"{name}: 256x256 persistent {ts[0]:.1f} us ({fl / ts[0] / 1e6:.0f} TF)  tile plan {ts[1]:.1f} us  hipBLASLt {ts[2]:.1f} us", flush=True)
