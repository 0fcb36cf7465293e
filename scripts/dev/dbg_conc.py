"""Do kernels from two processes run concurrently on one GPU?  Does the spin timeout fire?"""
import ctypes, os, sys, time, torch
sys.path.insert(0, "/root/repo")
import torch.distributed as dist
from triton_distributed_gh_amd.utils import initialize_distributed, finalize_distributed
from triton_distributed_gh_amd import _lib
env = initialize_distributed(heap_bytes=1 << 26)
r, w = env.rank, env.world_size
lib = _lib.hip()
heap = env.heap
flag = heap.create_tensor_list([16], torch.int32, arena="sig")
status = torch.zeros(2, dtype=torch.int64, device="cuda")
cs = heap.ctx_struct()
dist.barrier(group=env.cpu_group)
# test 1: timeout fires (nobody sets flag[1])
t0 = time.time()
_lib.check(lib.tdist_test_wait_flag(_lib.ptr(flag[r][1:]), 9, 2, _lib.ptr(status), _lib.stream_handle()))
torch.cuda.synchronize()
print(f"[r{r}] timeout test: status {status.tolist()} wall {time.time()-t0:.2f}s", flush=True)
dist.barrier(group=env.cpu_group)
# test 2: both ranks wait in a kernel for the peer's notify kernel, which is launched AFTER the wait kernel
s2 = torch.cuda.Stream()
t0 = time.time()
_lib.check(lib.tdist_test_wait_flag(_lib.ptr(flag[r]), 7, 2, _lib.ptr(status), _lib.stream_handle()))
time.sleep(0.2)
with torch.cuda.stream(s2):
    _lib.check(lib.tdist_test_notify(ctypes.byref(cs), _lib.ptr(flag[r]), (r + 1) % w, 7, 1, 2, _lib.stream_handle(s2)))
torch.cuda.synchronize()
print(f"[r{r}] cross-process kernel wait: status {status.tolist()} wall {time.time()-t0:.2f}s", flush=True)
dist.barrier(group=env.cpu_group)
finalize_distributed()
