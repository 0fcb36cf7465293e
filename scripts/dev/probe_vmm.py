"""Step-by-step probe of the HIP virtual-memory-management calls used by utils/vmm.py (one process)."""
import ctypes
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from triton_distributed_gh_amd import _lib  # noqa: E402

U = ctypes.c_uint64
torch.cuda.init()
torch.zeros(1, device="cuda")
lib = _lib.hip()
dev = torch.cuda.current_device()
for unc in (0, 1):
    g = U()
    rc = lib.tdist_vmm_granularity(dev, unc, ctypes.byref(g))
    print("granularity", unc, rc, g.value, flush=True)
chunk = 64 << 20
p = ctypes.c_void_p()
print("reserve", lib.tdist_vmm_reserve(ctypes.byref(p), U(4 * chunk), U(2 << 20)), hex(p.value or 0), flush=True)
h = U()
print("create", lib.tdist_vmm_create(dev, 0, U(chunk), ctypes.byref(h)), hex(h.value), flush=True)
print("map", lib.tdist_vmm_map(p, U(chunk), h, dev), flush=True)
print("memset", lib.tdist_memset(p, 0, ctypes.c_size_t(chunk)), flush=True)
fd = ctypes.c_int(-1)
print("export", lib.tdist_vmm_export_fd(h, ctypes.byref(fd)), fd.value, flush=True)
h2 = U()
print("import", lib.tdist_vmm_import_fd(fd.value, ctypes.byref(h2)), hex(h2.value), flush=True)
q = ctypes.c_void_p(p.value + chunk)
print("map2", lib.tdist_vmm_map(q, U(chunk), h2, dev), flush=True)
from triton_distributed_gh_amd.utils.symm import _wrap_device  # noqa: E402

a = _wrap_device(p.value, chunk, None)
b = _wrap_device(q.value, chunk, None)
a[:1024].fill_(7)
torch.cuda.synchronize()
print("alias sees", int(b[:1024].sum()), "(expected 7168)", flush=True)
print("last error:", lib.tdist_last_error(), flush=True)
