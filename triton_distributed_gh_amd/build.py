"""In-tree native build driver.

Compiles every HIP / C++ source under ``csrc/`` into shared libraries that live
next to this file:

* ``libtdist_hip.so``  - gfx950 device code + host launchers + hipIpc runtime
  (built with ``hipcc --offload-arch=gfx950``; cross-compiles without a GPU).
* ``libtdist_hip_safe.so`` - the same sources with ``-DTDIST_SAFE_PUBLISH=1``: every publish of a payload
  is followed by a system-scope release fence (``TDIST_SAFE_PUBLISH=1``, or the end of the heap self-test's
  fall-back chain, loads it instead).
* ``libtdist_host.so`` - CPU-only runtime (POSIX-shm symmetric heap emulation,
  signal wait/notify with std::atomic, CPU oracle of the AG-MoE tile swizzle).

The libraries expose a plain C ABI and are loaded with ``ctypes`` from
``_lib.py`` so rebuilding one kernel never recompiles torch headers.

Role in the reference: replaces ``python/setup.py`` + ``CMakeLists.txt`` +
``shmem/rocshmem_bind/build.sh`` (reference: python/setup.py:38-1050,
shmem/rocshmem_bind/build.sh:14-19) - there is no Triton/MLIR to build here.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ_DIR = ROOT / "build" / "obj"
ARCH = os.environ.get("TDIST_OFFLOAD_ARCH", "gfx950")
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))

HIP_LIB = ROOT / "libtdist_hip.so"
HIP_SAFE_LIB = ROOT / "libtdist_hip_safe.so"
HOST_LIB = ROOT / "libtdist_host.so"

HIPCC_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffast-math",
    "-fno-gpu-sanitize",
    "-Wno-unused-result",
    "-Wno-pass-failed",
    f"-I{CSRC / 'include'}",
    "-D__HIP_PLATFORM_AMD__",
]
if os.environ.get("TDIST_SPIN_TIMEOUT_MS"):
    # device-side spin budget in 100 MHz ticks (default in common.hpp: 15 s)
    HIPCC_FLAGS.append(f"-DTDIST_SPIN_TIMEOUT_TICKS={int(os.environ['TDIST_SPIN_TIMEOUT_MS']) * 100000}ull")
if os.environ.get("TDIST_EXTRA_HIPCC_FLAGS"):
    # e.g. -DTDIST_EXP_...=1 for an A/B build of one experiment (objects are keyed by their flags)
    HIPCC_FLAGS += os.environ["TDIST_EXTRA_HIPCC_FLAGS"].split()
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-pthread", f"-I{CSRC / 'include'}"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or str(ROCM / "bin" / "hipcc")
    return exe


def _digest(src: Path, flags: list[str]) -> str:
    h = hashlib.sha1()
    h.update(" ".join(flags).encode())
    h.update(src.read_bytes())
    for hdr in sorted((CSRC / "include").rglob("*.h*")):
        h.update(hdr.read_bytes())
    return h.hexdigest()[:16]


def _compile_one(cc: str, src: Path, flags: list[str], verbose: bool, prefix: str = "") -> Path:
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    obj = OBJ_DIR / f"{prefix}{src.stem}.{_digest(src, [cc] + flags)}.o"
    if obj.exists():
        return obj
    for stale in OBJ_DIR.glob(f"{prefix}{src.stem}.*.o"):
        stale.unlink()
    cmd = [cc, *flags, "-c", str(src), "-o", str(obj)]
    if verbose:
        print("[tdist build]", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError(f"compile failed: {src}")
    if verbose and res.stderr.strip():
        sys.stderr.write(res.stderr)
    return obj


def hip_sources() -> list[Path]:
    return sorted((CSRC / "ops").glob("*.hip")) + sorted((CSRC / "runtime").glob("*_hip.cc"))


def host_sources() -> list[Path]:
    return sorted(p for p in (CSRC / "runtime").glob("*.cc") if not p.name.endswith("_hip.cc"))


def build_hip(verbose: bool = True, jobs: int | None = None, safe: bool = False) -> Path:
    """``safe``: the TDIST_SAFE_PUBLISH=1 flavour (own objects, own library)."""
    cc = _hipcc()
    srcs = hip_sources()
    jobs = jobs or min(8, max(1, (os.cpu_count() or 2)))
    flags = list(HIPCC_FLAGS) + (["-DTDIST_SAFE_PUBLISH=1"] if safe else [])
    lib = HIP_SAFE_LIB if safe else HIP_LIB
    prefix = "safe-" if safe else ""
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile_one(cc, s, flags, verbose, prefix), srcs))
    newest = max(o.stat().st_mtime for o in objs)
    if lib.exists() and lib.stat().st_mtime >= newest:
        return lib
    # -Bsymbolic: the library's own calls bind to itself, so both flavours can live in one process
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-sanitize", "-Wl,-Bsymbolic", *map(str, objs), "-o",
           str(lib)]
    if verbose:
        print("[tdist build] link", lib.name, flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError(f"link failed: {lib.name}")
    return lib


def build_hip_safe(verbose: bool = True, jobs: int | None = None) -> Path:
    return build_hip(verbose, jobs, safe=True)


def build_host(verbose: bool = True) -> Path:
    cc = shutil.which("g++") or "g++"
    srcs = host_sources()
    objs = [_compile_one(cc, s, HOST_FLAGS, verbose) for s in srcs]
    newest = max(o.stat().st_mtime for o in objs)
    if HOST_LIB.exists() and HOST_LIB.stat().st_mtime >= newest:
        return HOST_LIB
    cmd = [cc, "-shared", "-fPIC", "-pthread", *map(str, objs), "-o", str(HOST_LIB), "-lrt"]
    if verbose:
        print("[tdist build] link", HOST_LIB.name, flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("link failed: libtdist_host.so")
    return HOST_LIB


def build_all(verbose: bool = True) -> None:
    build_host(verbose)
    build_hip(verbose)
    build_hip(verbose, safe=True)


if __name__ == "__main__":
    build_all(verbose=True)
    print("ok:", HIP_LIB, HIP_SAFE_LIB, HOST_LIB)
