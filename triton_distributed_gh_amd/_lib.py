"""ctypes loaders for the two in-tree native libraries.

``hip()``  -> libtdist_hip.so  (gfx950 kernels + hipIpc runtime; needs a GPU process), or libtdist_hip_safe.so -
             the same code with a system-scope release fence behind every payload publish - when
             ``TDIST_SAFE_PUBLISH=1`` is set or ``use_safe_publish()`` was called (the symmetric heap's self-test
             does that when the zero-fence publish fails in every memory mode)
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


def _load(path: Path, builder, mode: int = ctypes.RTLD_GLOBAL) -> ctypes.CDLL:
    if not path.exists():
        if os.environ.get("TDIST_NO_AUTOBUILD", "0") == "1":
            raise NativeError(f"{path.name} is missing; run `python -m triton_distributed_gh_amd.build`")
        builder(verbose=False)
    return ctypes.CDLL(str(path), mode=mode)


_HIP = {"lib": None, "safe": False}


def _load_hip(safe: bool) -> ctypes.CDLL:
    from . import build

    if safe:
        # local + deep binding: this flavour's symbols must not be interposed by the default one (if loaded)
        lib = _load(build.HIP_SAFE_LIB, build.build_hip_safe, mode=os.RTLD_LOCAL | os.RTLD_NOW | os.RTLD_DEEPBIND)
    else:
        lib = _load(build.HIP_LIB, build.build_hip)
    lib.tdist_last_error.restype = ctypes.c_char_p
    if bool(lib.tdist_build_safe_publish()) != safe:
        raise NativeError(f"{'libtdist_hip_safe.so' if safe else 'libtdist_hip.so'} was built with the wrong "
                          "TDIST_SAFE_PUBLISH setting; rebuild with `python -m triton_distributed_gh_amd.build`")
    return lib


def hip() -> ctypes.CDLL:
    lib = _HIP["lib"]
    if lib is None:
        _HIP["safe"] = os.environ.get("TDIST_SAFE_PUBLISH", "0") == "1"
        lib = _HIP["lib"] = _load_hip(_HIP["safe"])
    return lib


def safe_publish() -> bool:
    """True when the loaded GPU library is the release-fence flavour."""
    hip()
    return _HIP["safe"]


def use_safe_publish() -> None:
    """Switch this process to the release-fence flavour (every later native call goes to it).  Contexts made
    before the switch stay valid: they hold device memory and plain integers, no function pointers."""
    if _HIP["lib"] is not None and _HIP["safe"]:
        return
    _HIP["lib"] = _load_hip(True)
    _HIP["safe"] = True


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
