"""``extern_call``: call an exported native entry point by symbol name
(reference: python/triton_dist/language/core.py:85-117 emits a call to an external device symbol;
the native library here exposes host launchers with a C ABI, which is what can be called by name)."""
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
