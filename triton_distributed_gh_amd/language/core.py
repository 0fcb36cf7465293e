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


def _type_key(a):
    import torch

    if isinstance(a, torch.Tensor):
        return a.dtype
    if isinstance(a, ctypes._SimpleCData):
        return type(a)
    return type(a)


def dispatch(func, lib_name: str, lib_path: str, args: list, arg_type_symbol_dict: dict, is_pure: bool = False, **_):
    """Pick the symbol whose key in ``arg_type_symbol_dict`` equals the tuple of argument types (torch dtypes for
    tensors, the Python / ctypes type otherwise) and call it: ``func(symbol, args)`` when ``func`` is given,
    ``extern_call(symbol, args, lib=lib_name)`` otherwise.  ``lib_path`` names a shared library to load instead of
    the package's own.  (Reference: python/triton_dist/language/core.py:33-83 resolves an overloaded external
    DEVICE symbol this way at Triton compile time; here it is the host-side overload resolution.)"""
    key = tuple(_type_key(a) for a in args)
    if key not in arg_type_symbol_dict:
        raise ValueError(f"no overload for argument types {key}; known: {sorted(map(str, arg_type_symbol_dict))}")
    entry = arg_type_symbol_dict[key]
    symbol, restype = entry if isinstance(entry, tuple) else (entry, ctypes.c_int)
    if func is not None:
        return func(symbol, args)
    if lib_path:
        fn = getattr(ctypes.CDLL(lib_path), symbol)
        fn.restype = restype
        return fn(*args)
    return extern_call(symbol, args, restype=restype, lib=lib_name or "hip", check=False)


def extern_elementwise(lib_name: str, lib_path: str, args: list, arg_type_symbol_dict: dict, is_pure: bool = False, **kw):
    """Overload-resolved call of an external element-wise launcher (tensor arguments pass as pointers): the host-side
    counterpart of the reference's pointer-capable ``extern_elementwise`` (python/triton_dist/language/core.py:121-164).
    The symbol receives the arguments as given, tensors as ``void*``."""
    import torch

    conv = [ctypes.c_void_p(a.data_ptr()) if isinstance(a, torch.Tensor) else a for a in args]
    key = tuple(_type_key(a) for a in args)
    if key not in arg_type_symbol_dict:
        raise ValueError(f"no overload for argument types {key}")
    return dispatch(None, lib_name, lib_path, conv, {tuple(_type_key(a) for a in conv): arg_type_symbol_dict[key]}, is_pure)
