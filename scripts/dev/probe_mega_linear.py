import sys, torch
sys.path.insert(0, "/root/repo")
import triton_distributed_gh_amd.mega.model_builder as mb
from triton_distributed_gh_amd.mega import ModelBuilder
from triton_distributed_gh_amd.utils.perf import perf_func
BF = torch.bfloat16
N, K, M = 65536, 8192, 1
x = torch.randn(M, K, device="cuda").to(BF)
w = torch.randn(N, K, device="cuda").to(BF)
out = torch.empty(M, N, device="cuda", dtype=BF)
gb = N * K * 2 / 1e9
_, t = perf_func(lambda: torch.matmul(x, w.t()), 20, 5)
print("torch.matmul  %.1f us  %.2f TB/s" % (t * 1e3, gb / t), flush=True)
for queues in (256, 512, 1024):
    for bn in (16, 32, 64):
        mb._linear_bn = lambda N_, q, bn=bn: bn
        b = ModelBuilder(num_queues=queues)
        b.make_linear(x, w, out)
        b.compile(validate=False)
        _, t = perf_func(b.run, 20, 5)
        b.check_error()
        print("queues %4d BN %2d  %.1f us  %.2f TB/s" % (queues, bn, t * 1e3, gb / t), flush=True)
# dependency chain overhead: 64 small dependent ops
b = ModelBuilder(num_queues=256)
h = torch.randn(1, 4096, device="cuda").to(BF)
wn = torch.ones(4096, device="cuda", dtype=BF)
bufs = [torch.empty(1, 4096, device="cuda", dtype=BF) for _ in range(2)]
cur = h
for i in range(200):
    b.make_rms_norm(cur, wn, bufs[i % 2])
    cur = bufs[i % 2]
b.compile(validate=False)
_, t = perf_func(b.run, 20, 5)
print("200 dependent 1-tile norm tasks: %.1f us  (%.2f us per hop)" % (t * 1e3, t * 1e3 / 200), flush=True)
