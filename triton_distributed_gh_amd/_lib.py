"""ctypes loaders for the two in-tree native libraries.

``hip()``  -> libtdist_hip.so  (gfx950 kernels + hipIpc runtime; needs a GPU process)
``host()`` -> libtdist_host.so (CPU runtime: shm symmetric heap, atomics-based signals)

Both libraries expose a C ABI where every function returns 0 on success; ``check`` turns a
non-zero code into a Python exception carrying the library's last error string.
"""
from __future__ import annotations

import ctypes
import functools
import os
from pathlib import Path

_ROOT = Path(__file__).resolve().parent


class NativeError(RuntimeError):
    pass


def _load(path: Path, builder) -> ctypes.CDLL:
    if not path.exists():
        if os.environ.get("TDIST_NO_AUTOBUILD", "0") == "1":
            raise NativeError(f"{path.name} is missing; run `python -m triton_distributed_gh_amd.build`")
        builder(verbose=False)
    return ctypes.CDLL(str(path), mode=ctypes.RTLD_GLOBAL)


@functools.lru_cache(maxsize=None)
def hip() -> ctypes.CDLL:
    from . import build

    lib = _load(build.HIP_LIB, build.build_hip)
    lib.tdist_last_error.restype = ctypes.c_char_p
    return lib


@functools.lru_cache(maxsize=None)
def host() -> ctypes.CDLL:
    from . import build

    lib = _load(build.HOST_LIB, build.build_host)
    lib.tdh_last_error.restype = ctypes.c_char_p
    lib.tdh_load32.restype = ctypes.c_int32
    lib.tdh_load64.restype = ctypes.c_int64
    return lib


def hip_available() -> bool:
    """True when this process can see a GPU (without initialising HIP through torch twice)."""
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = hip().tdist_last_error().decode(errors="replace")
        raise NativeError(f"{what or 'native call'} failed with code {code}: {msg}")


def check_host(code: int, what: str = "") -> None:
    if code != 0:
        msg = host().tdh_last_error().decode(errors="replace")
        raise NativeError(f"{what or 'host call'} failed with code {code}: {msg}")


def ptr(t) -> ctypes.c_void_p:
    """Raw device/host pointer of a tensor (or an int address) as a ctypes void*."""
    if t is None:
        return ctypes.c_void_p(0)
    if isinstance(t, int):
        return ctypes.c_void_p(t)
    return ctypes.c_void_p(t.data_ptr())


def stream_handle(stream=None) -> ctypes.c_void_p:
    import torch

    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)
