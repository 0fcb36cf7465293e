"""Two-process probe of dmabuf export / import of a hipMemCreate chunk (parent exports, child imports)."""
import ctypes
import faulthandler
import os
import socket
import subprocess
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402

from triton_distributed_gh_amd import _lib  # noqa: E402

U = ctypes.c_uint64
ADDR = "\0tdist_probe_vmm_ipc"
chunk = 64 << 20
role = sys.argv[1] if len(sys.argv) > 1 else "parent"
torch.zeros(1, device="cuda")
lib = _lib.hip()
dev = torch.cuda.current_device()
if role == "parent":
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind(ADDR)
    srv.listen(1)
    p = ctypes.c_void_p()
    print("reserve", lib.tdist_vmm_reserve(ctypes.byref(p), U(chunk), U(2 << 20)), flush=True)
    h = U()
    print("create", lib.tdist_vmm_create(dev, 0, U(chunk), ctypes.byref(h)), flush=True)
    print("map", lib.tdist_vmm_map(p, U(chunk), h, dev), flush=True)
    from triton_distributed_gh_amd.utils.symm import _wrap_device
    a = _wrap_device(p.value, chunk, None)
    a.fill_(5)
    torch.cuda.synchronize()
    fd = ctypes.c_int(-1)
    print("export", lib.tdist_vmm_export_fd(h, ctypes.byref(fd)), fd.value, os.readlink(f"/proc/self/fd/{fd.value}"), flush=True)
    env = dict(os.environ)
    if len(sys.argv) > 2:
        env["AMD_LOG_LEVEL"] = sys.argv[2]
    child = subprocess.Popen([sys.executable, "-u", __file__, "child"], env=env)
    conn, _ = srv.accept()
    socket.send_fds(conn, [b"x"], [fd.value])
    print("parent: child says", conn.recv(64), flush=True)
    child.wait()
    print("parent sees", int(a[:4].sum()), "(child wrote 9s -> 36)", flush=True)
else:
    c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    c.connect(ADDR)
    msg, fds, _, _ = socket.recv_fds(c, 1, 1)
    print("child fd", fds, os.readlink(f"/proc/self/fd/{fds[0]}"), flush=True)
    h2 = U()
    rc = lib.tdist_vmm_import_fd(fds[0], ctypes.byref(h2))
    print("child import", rc, lib.tdist_last_error(), flush=True)
    if rc == 0:
        q = ctypes.c_void_p()
        print("child reserve", lib.tdist_vmm_reserve(ctypes.byref(q), U(chunk), U(2 << 20)), flush=True)
        print("child map", lib.tdist_vmm_map(q, U(chunk), h2, dev), lib.tdist_last_error(), flush=True)
        from triton_distributed_gh_amd.utils.symm import _wrap_device
        b = _wrap_device(q.value, chunk, None)
        print("child sees", int(b[:4].sum()), "(expected 20)", flush=True)
        b.fill_(9)
        torch.cuda.synchronize()
    c.send(b"done")
