"""Minimal two-process hipIpc probe (run through scripts/dev/mp_debug.sh): which allocation kinds
and sizes can be exported / opened between two processes on this machine, and how long it takes."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402

from triton_distributed_gh_amd import _lib  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
tag = os.environ.get("MASTER_PORT", "0")
ndev = torch.cuda.device_count()
torch.cuda.set_device(rank % ndev)
lib = _lib.hip()
hsz = lib.tdist_ipc_handle_size()


def log(*a):
    print(f"[rank {rank} t={time.time() % 1000:8.3f}]", *a, flush=True)


def exchange(name, payload: bytes) -> bytes:
    path = f"/tmp/ipc_{tag}_{name}_{rank}"
    with open(path + ".tmp", "wb") as f:
        f.write(payload)
    os.rename(path + ".tmp", path)
    peer = f"/tmp/ipc_{tag}_{name}_{(rank + 1) % world}"
    while not os.path.exists(peer):
        time.sleep(0.01)
    return open(peer, "rb").read()


log(f"devices={ndev} HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}")
cases = [("hipMalloc 1MiB", 1 << 20, 0), ("finegrained 1MiB", 1 << 20, 1), ("hipMalloc 256MiB", 1 << 28, 0),
         ("hipMalloc 2GiB", 1 << 31, 0)]
if os.environ.get("IPC_SIZES"):
    cases = [(f"hipMalloc {int(x, 0)} B", int(x, 0), 0) for x in os.environ["IPC_SIZES"].split(",")]
only = os.environ.get("IPC_CASE")
for i, (name, nbytes, mode) in enumerate(cases):
    if only is not None and str(i) != only:
        continue
    p = ctypes.c_void_p()
    _lib.check(lib.tdist_malloc(ctypes.byref(p), ctypes.c_size_t(nbytes), mode), "malloc")
    h = ctypes.create_string_buffer(hsz)
    t0 = time.time()
    _lib.check(lib.tdist_ipc_get_handle(p, h), "get_handle")
    log(f"{name}: exported in {time.time() - t0:.3f}s")
    peer_h = exchange(f"h{i}", bytes(h.raw))
    if os.environ.get("IPC_SERIAL") == "1" and rank == 1:
        exchange(f"turn{i}", b"x")            # rank 1 opens only after rank 0 is done
    q = ctypes.c_void_p()
    t0 = time.time()
    log(f"{name}: opening peer handle ...")
    rc = lib.tdist_ipc_open_handle(peer_h, ctypes.byref(q))
    log(f"{name}: open rc={rc} ptr={q.value} in {time.time() - t0:.3f}s")
    if os.environ.get("IPC_SERIAL") == "1" and rank == 0:
        exchange(f"turn{i}", b"x")
    exchange(f"done{i}", b"x")
log("all cases done")
