"""GPU probe: multi-process symmetric heap on ONE device (hipIpc), copy engine, stream values."""
import ctypes, os, sys, time, torch
sys.path.insert(0, "/root/repo")
import torch.distributed as dist
from triton_distributed_gh_amd.utils import initialize_distributed, finalize_distributed
from triton_distributed_gh_amd import _lib

env = initialize_distributed(heap_bytes=1 << 28)
r, w = env.rank, env.world_size
print(f"rank {r}/{w} dev {env.device} shared_gpu {env.shared_gpu} rccl {env.has_rccl}", flush=True)
heap = env.heap
ts = heap.create_tensor_list([w, 1024], torch.float32)
sig = heap.create_tensor_list([64], torch.int32, arena="sig")
dist.barrier(group=env.cpu_group)
# 1. plain peer store through torch
for p in range(w):
    ts[p][r].fill_(float(r + 1))
torch.cuda.synchronize(); dist.barrier(group=env.cpu_group)
ok = all(float(ts[r][s][0]) == s + 1 for s in range(w))
print(f"rank {r} peer stores visible: {ok}", flush=True)
# 2. copy engine
lib = _lib.hip()
src = torch.full((1024,), 100.0 + r, device="cuda")
st = torch.cuda.Stream()
for p in range(w):
    dst_ptr = heap.peer_ptr(ts[r], p) + r * 1024 * 4
    _lib.check(lib.tdist_memcpy_async(ctypes.c_void_p(dst_ptr), _lib.ptr(src), ctypes.c_size_t(4096), 1, _lib.stream_handle(st)), "memcpy nocu")
    _lib.check(lib.tdist_stream_write_value32(_lib.stream_handle(st), ctypes.c_void_p(heap.peer_ptr(sig[r], p) + 4 * r), 7), "write value")
st.synchronize(); dist.barrier(group=env.cpu_group)
ok = all(float(ts[r][s][5]) == 100.0 + s for s in range(w)) and all(int(sig[r][s]) == 7 for s in range(w))
print(f"rank {r} copy-engine + write_value visible: {ok}  sig={sig[r][:w].tolist()}", flush=True)
# 3. wait value on stream: peer sets, we wait
_lib.check(lib.tdist_stream_wait_value32(_lib.stream_handle(st), _lib.ptr(sig[r][32:]), 5, 0), "wait value")
time.sleep(0.2 * r)
nxt = (r + 1) % w
_lib.check(lib.tdist_stream_write_value32(_lib.stream_handle(), ctypes.c_void_p(heap.peer_ptr(sig[r], nxt) + 4 * 32), 5), "write value2")
st.synchronize()
print(f"rank {r} stream wait_value released", flush=True)
dist.barrier(group=env.cpu_group)
finalize_distributed()
print(f"rank {r} done", flush=True)
