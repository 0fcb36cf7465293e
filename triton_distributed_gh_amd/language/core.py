"""``extern_call``: call an exported native entry point by symbol name.

The reference's ``extern_call`` (python/triton_dist/language/core.py:85-117) makes a Triton kernel call a
symbol of an external DEVICE library.  Kernels here are HIP source, so the device-side form is the language
itself: the kernel file declares ``extern "C" __device__ T sym(...)`` and ``triton_dist.tools.compile_hip(...,
extern_libs=[lib.hip])`` links the library's relocatable device code into the kernel's code object
(tools/runtime.py; tests/test_user_kernels.py builds and runs one).  The function below is the HOST-side
counterpart: any launcher the native libraries export with a C ABI, called by name."""
from __future__ import annotations

import ctypes

from .. import _lib


def extern_call(symbol: str, args=(), restype=ctypes.c_int, lib: str = "hip", check: bool = True):
    handle = _lib.hip() if lib == "hip" else _lib.host()
    try:
        fn = getattr(handle, symbol)
    except AttributeError as e:
        raise AttributeError(f"native symbol {symbol!r} is not exported by libtdist_{lib}.so") from e
    fn.restype = restype
    rc = fn(*args)
    if check and restype is ctypes.c_int and rc != 0:
        (_lib.check if lib == "hip" else _lib.check_host)(rc, symbol)
    return rc
