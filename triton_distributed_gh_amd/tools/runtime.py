"""User kernels: compile a stand-alone ``.hip`` file against the device headers, load the code object and
launch its kernels from Python with symmetric-heap arguments.

    from triton_dist.tools import compile_hip, load_module
    mod = load_module(compile_hip("my_kernel.hip"))              # hipcc --genco, gfx950
    k = mod.get("my_kernel")                                      # extern "C" __global__ void my_kernel(...)
    k(grid, block, stream, env.heap, queue, signal, 42, f32(0.5))   # SymmCtx by value, tensors as pointers

This is the path the reference serves with its JIT (``@triton.jit`` kernels calling ``dl.wait`` /
``dl.notify`` / ``dl.symm_at``, reference: python/triton_dist/language/distributed_ops.py:56-111,
tutorials/01-distributed-notify-wait.py:62-145) and, ahead of time, with ``compile_aot`` + the C++ module
loader (tools/compile_aot.py:204-295, tools/runtime/triton_aot_runtime.cc:54-199).  Here a user kernel is
plain HIP that includes ``tdist/primitives.hpp`` (wait / notify / consume_token / symm_at / rank /
num_ranks) and, if wanted, ``tdist/shmem_device.hpp``; the loader is csrc/runtime/aot_runtime_hip.cc
(hipModuleLoad / hipModuleGetFunction / hipModuleLaunchKernel behind a per-device cache).

Argument conversion (one kernel parameter each, in order):
    torch.Tensor        -> its data pointer (``T*``)
    SymmetricHeap       -> ``tdist::SymmCtx`` by value (the heap's current peer table)
    ctypes instance     -> passed as is (any struct or explicitly sized scalar)
    bool / int          -> ``int`` (32 bit);  use ``i64(x)`` / ``u64(x)`` for 64-bit parameters
    float               -> ``float``;         use ``f64(x)`` for double
    None                -> null pointer
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import shutil
import subprocess
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

from .. import _lib

INCLUDE_DIR = Path(__file__).resolve().parent.parent / "csrc" / "include"


def i32(x) -> ctypes.c_int:
    return ctypes.c_int(int(x))


def i64(x) -> ctypes.c_longlong:
    return ctypes.c_longlong(int(x))


def u64(x) -> ctypes.c_ulonglong:
    return ctypes.c_ulonglong(int(x))


def f32(x) -> ctypes.c_float:
    return ctypes.c_float(float(x))


def f64(x) -> ctypes.c_double:
    return ctypes.c_double(float(x))


def hipcc_path() -> Optional[str]:
    return shutil.which("hipcc") or (str(Path(os.environ.get("ROCM_PATH", "/opt/rocm")) / "bin" / "hipcc")
                                     if (Path(os.environ.get("ROCM_PATH", "/opt/rocm")) / "bin" / "hipcc").exists() else None)


def _llvm_tool(name: str) -> Optional[str]:
    p = Path(os.environ.get("ROCM_PATH", "/opt/rocm")) / "lib" / "llvm" / "bin" / name
    return str(p) if p.exists() else shutil.which(name)


def compile_hip(source: Union[str, Path], out: Union[str, Path, None] = None, arch: str = "gfx950",
                extra_flags: Sequence[str] = (), force: bool = False,
                extern_libs: Sequence[Union[str, Path]] = ()) -> Path:
    """``hipcc --genco`` of one source file with the device headers on the include path.  The code
    object is rebuilt only when the source, the headers or the flags changed (a digest is kept next to
    it), so calling this at every start-up is cheap.

    ``extern_libs``: more ``.hip`` sources that DEFINE device functions the kernel file only declares
    (``extern "C" __device__ float user_act(float);``).  Every file is compiled to relocatable device bitcode
    (``-fgpu-rdc``), the bitcode is linked and one code object is produced - the device-side form of the
    reference's ``extern_call`` / ``extern_libs`` (python/triton_dist/language/core.py:85-117: a Triton kernel
    calling a symbol of an external device library)."""
    src = Path(source).resolve()
    libs = [Path(p).resolve() for p in extern_libs]
    out = Path(out) if out is not None else src.with_suffix(".hsaco")
    common = [f"--offload-arch={arch}", "-O3", "-std=c++17", "-fno-gpu-sanitize", f"-I{INCLUDE_DIR}", *extra_flags]
    flags = (["--genco"] if not libs else ["-fgpu-rdc", "--cuda-device-only", "-c"]) + common
    h = hashlib.sha1(" ".join(flags).encode())
    for f in [src, *libs]:
        h.update(f.read_bytes())
    for hdr in sorted(INCLUDE_DIR.rglob("*.h*")):
        h.update(hdr.read_bytes())
    digest = h.hexdigest()
    stamp = out.with_suffix(out.suffix + ".sha1")
    if not force and out.exists() and stamp.exists() and stamp.read_text().strip() == digest:
        return out
    cc = hipcc_path()
    if cc is None:
        if out.exists():
            return out                      # no compiler on this machine: use the code object that shipped
        raise RuntimeError("hipcc not found and no prebuilt code object at " + str(out))
    out.parent.mkdir(parents=True, exist_ok=True)

    def run(cmd, what):
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"{what} failed for {src}:\n{res.stdout}{res.stderr}")

    if not libs:
        run([cc, *flags, str(src), "-o", str(out)], "hipcc")
    else:
        link, clang = _llvm_tool("llvm-link"), _llvm_tool("clang")
        if link is None or clang is None:
            raise RuntimeError("extern_libs needs llvm-link and clang of the ROCm toolchain")
        bcs = []
        for k, f in enumerate([src, *libs]):
            bc = out.with_suffix(f".{k}.bc")
            run([cc, *flags, str(f), "-o", str(bc)], "hipcc (device bitcode)")
            bcs.append(str(bc))
        merged = out.with_suffix(".linked.bc")
        run([link, *bcs, "-o", str(merged)], "llvm-link")
        run([clang, "-target", "amdgcn-amd-amdhsa", f"-mcpu={arch}", "-O3", str(merged), "-o", str(out)], "device link")
        for f in [*bcs, str(merged)]:
            os.unlink(f)
    stamp.write_text(digest)
    return out


def marshal_args(args: Sequence) -> Tuple[ctypes.Array, List]:
    """Kernel arguments in the hipModuleLaunchKernel convention: an array of pointers, element i
    pointing at the value of parameter i.  Returns (array, keep-alive list of the values)."""
    import torch

    from ..utils.symm import SymmetricHeap

    vals: List = []
    for a in args:
        if a is None:
            vals.append(ctypes.c_void_p(0))
        elif isinstance(a, torch.Tensor):
            vals.append(ctypes.c_void_p(a.data_ptr()))
        elif isinstance(a, SymmetricHeap):
            vals.append(a.ctx_struct())
        elif isinstance(a, (ctypes._SimpleCData, ctypes.Structure, ctypes.Array)):
            vals.append(a)
        elif isinstance(a, bool):
            vals.append(ctypes.c_int(int(a)))
        elif isinstance(a, int):
            if not -(1 << 31) <= a < (1 << 31):
                raise TypeError(f"integer argument {a} does not fit an int; wrap it in i64() / u64()")
            vals.append(ctypes.c_int(a))
        elif isinstance(a, float):
            vals.append(ctypes.c_float(a))
        else:
            raise TypeError(f"cannot pass {type(a).__name__} to a kernel; use a tensor, a heap, a number or a ctypes value")
    arr = (ctypes.c_void_p * max(len(vals), 1))(*[ctypes.addressof(v) for v in vals])
    return arr, vals


def _dim3(x) -> Tuple[int, int, int]:
    if isinstance(x, int):
        return (x, 1, 1)
    t = tuple(int(v) for v in x) + (1, 1, 1)
    return t[:3]


class Kernel:
    def __init__(self, module: "Module", name: str, handle: int):
        self.module, self.name, self._fn = module, name, ctypes.c_void_p(handle)
        self._max_lds = 0

    def __call__(self, grid, block, stream, *args, lds_bytes: int = 0) -> None:
        """Launch on ``stream`` (a torch stream, a raw handle or None for the current stream)."""
        g, b = _dim3(grid), _dim3(block)
        if b[0] * b[1] * b[2] > 1024 or min(g + b) < 1:
            raise ValueError(f"bad launch shape grid={g} block={b}")
        lib = _lib.hip()
        if lds_bytes > 65536 and lds_bytes > self._max_lds:
            _lib.check(lib.tdist_aot_func_set_max_dynamic_lds(self._fn, int(lds_bytes)), "set max dynamic LDS")
            self._max_lds = lds_bytes
        arr, keep = marshal_args(args)
        if stream is None or hasattr(stream, "cuda_stream"):
            sh = _lib.stream_handle(stream)
        else:
            sh = ctypes.c_void_p(int(stream))
        _lib.check(lib.tdist_aot_launch(self._fn, g[0], g[1], g[2], b[0], b[1], b[2], int(lds_bytes), sh, arr),
                   f"launch of {self.name}")
        del keep        # hipModuleLaunchKernel copied the values into the kernarg segment


    def max_active_blocks_per_cu(self, block_threads: int, lds_bytes: int = 0) -> int:
        """Workgroups of this kernel one compute unit keeps resident at the given block size and dynamic LDS
        (reference: python/triton_dist/utils.py ``cuda_occupancy_max_activate_blocks_per_multiprocessor``).
        A kernel whose workgroups wait for each other must not launch more than this times the CU count."""
        out = ctypes.c_int(0)
        _lib.check(_lib.hip().tdist_aot_occupancy(self._fn, int(block_threads), int(lds_bytes), ctypes.byref(out)),
                   f"occupancy of {self.name}")
        return int(out.value)


class Module:
    """A loaded code object.  ``get(name)`` returns a launchable kernel (``extern "C"`` name)."""

    def __init__(self, path: Union[str, Path]):
        self.path = str(path)
        h = ctypes.c_void_p()
        _lib.check(_lib.hip().tdist_aot_module_load_file(self.path.encode(), ctypes.byref(h)), f"load {self.path}")
        self._handle = h
        self._kernels: Dict[str, Kernel] = {}

    def get(self, name: str) -> Kernel:
        k = self._kernels.get(name)
        if k is None:
            f = ctypes.c_void_p()
            _lib.check(_lib.hip().tdist_aot_module_get_function(self._handle, name.encode(), ctypes.byref(f)),
                       f"kernel {name!r} in {self.path} (declare it extern \"C\")")
            k = self._kernels[name] = Kernel(self, name, f.value)
        return k

    __getitem__ = get

    def unload(self) -> None:
        if self._handle is not None:
            _lib.check(_lib.hip().tdist_aot_module_unload(self._handle), "module unload")
            self._handle, self._kernels = None, {}


_MODULES: Dict[str, Module] = {}


def load_module(path: Union[str, Path]) -> Module:
    """Load (once per process and path) a ``.hsaco`` produced by ``compile_hip`` / ``compile_aot --build``."""
    key = str(Path(path).resolve())
    m = _MODULES.get(key)
    if m is None or m._handle is None:
        m = _MODULES[key] = Module(key)
    return m


def load_workspace(workspace: Union[str, Path]) -> Dict[str, Kernel]:
    """Kernels of every code object listed in a ``compile_aot --build`` workspace manifest, by kernel name."""
    import json

    man = json.loads((Path(workspace) / "manifest.json").read_text())
    out = {}
    for co in man.get("code_objects", []):
        out[co["kernel"]] = load_module(co["hsaco"]).get(co["kernel"])
    return out
