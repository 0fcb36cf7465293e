"""Compute-side cost of fast_all_to_all for ONE rank: rank 0 of an 8-rank virtual cluster runs alone
on the GPU with its arrival signals pre-set, so nothing waits and every put lands in local HBM.
DeepSeek decode shape: 128 tokens x topk 8 rows per rank, hidden 7168, fp8 + per-128 scales."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from triton_distributed_gh_amd.ops.all_to_all import create_all_to_all_context, fast_all_to_all, quantize_fp8_groupwise
from triton_distributed_gh_amd.utils.testing import VirtualCluster

w, G, rows, hidden = 8, 32, 1024, 7168
E = w * G
vc = VirtualCluster(w, heap_bytes=512 << 20)
ctxs = vc.collective(lambda r, env: create_all_to_all_context(rows, hidden, r, E, w, G, torch.float8_e4m3fn, env=env))
g = torch.Generator(device="cpu").manual_seed(0)
dest = torch.sort(torch.randint(0, E, (rows,), generator=g)).values
cum = torch.zeros(E + 1, dtype=torch.int32)
cum[1:] = torch.cumsum(torch.bincount(dest, minlength=E), 0)
cum = cum.cuda()
x, s = quantize_fp8_groupwise(torch.randn(rows, hidden, generator=g).cuda().bfloat16())
ctx = ctxs[0]
for slices in (1, 2, 4, 8, 16):
    os.environ["TDIST_A2A_SLICES"] = str(slices)
    times = []
    for it in range(25):
        ctx.signal_buf[0].fill_(int(ctx.state[0]) % 4093 + 1)      # every peer "has arrived" for this call (device-numbered)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fast_all_to_all(ctx, x, cum, s)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3)
    times = sorted(times[5:])
    nbytes = rows * hidden + rows * (hidden // 128) * 4
    print(f"slices={slices:2d} (grid {w * slices:3d}): median {times[len(times) // 2]:.1f} us, min {times[0]:.1f} us, "
          f"{nbytes / times[len(times) // 2] / 1e3:.1f} GB/s", flush=True)
ctx.check_error()
vc.close()
