"""Compute-side latency of all_reduce for ONE rank: rank 0 of an 8-rank virtual cluster runs alone with
its peers' flags pre-set (nothing waits; pushes land in local HBM).  HIP events, median of 20."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from triton_distributed_gh_amd.ops.allreduce import (AllReduceMethod, EPOCH_MOD, all_reduce, create_allreduce_ctx,
                                                     get_auto_allreduce_method)
from triton_distributed_gh_amd.utils.testing import VirtualCluster

w = 8
vc = VirtualCluster(w, heap_bytes=512 << 20)
ctxs = vc.collective(lambda r, env: create_allreduce_ctx(64 << 20, r, w, env=env))
c = ctxs[0]
nbytes = 4096
while nbytes <= (16 << 20):
    x = (torch.rand(nbytes // 2, device="cuda") - 0.5).bfloat16()
    out = torch.empty_like(x)
    for method in (AllReduceMethod.OneShot, AllReduceMethod.TwoShot):
        if method == AllReduceMethod.OneShot and nbytes > (4 << 20):
            continue
        ts = []
        for it in range(24):
            c.flags[c.call_count & 1][0].fill_(c.call_count % EPOCH_MOD + 1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); all_reduce(x, method, c, output=out); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[4:])
        auto = "  <- auto" if get_auto_allreduce_method(nbytes) == method else ""
        print(f"nbytes {nbytes:>9} {method.name:>8}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us{auto}", flush=True)
    nbytes *= 4
c.check_error()
vc.close()
