"""AOT kernel registry: specs -> native dispatch table -> ``distributed.<kernel>(stream, args..., algo_info)``.

Reference: python/triton_dist/tools/compile_aot.py (``aot_compile_spaces`` :61-77, algo-info struct
generation :187-201, pybind with algo-info :357-435, ``link_all`` :470-679, CLI :758-843) compiles
Triton kernels to cubins and generates C launchers + a pybind module.  Every kernel of this
framework is ALREADY compiled ahead of time (hipcc, gfx950) into libtdist_hip.so, so "AOT" reduces
to what remains useful:

* the SPEC FORMAT is kept: ``{c_kernel_name: {"signature": "*fp16:16, *fp32, i32:1, fp32, %BLOCK",
  "grid": ["expr_x", "expr_y", "expr_z"], "triton_algo_infos": [{...}, ...]}}`` with ``*T[:align]``
  pointers, ``i32[:1|:16]`` hints and ``%NAME`` constants taken from the algo-info;
* specs are validated and registered in the native registry (csrc/runtime/registry.cc), which also
  picks the algo-info variant at call time;
* each kernel name becomes ``distributed.<name>(stream, args..., algo_info)`` plus the settable
  struct ``distributed.<name>__triton_algo_info_t`` - the reference's calling convention
  (test/nvidia/test_compile_aot.py:174-182) - bound to a native launcher;
* the CLI (``--workspace --from-scratch --kernels path:kernel ... --library NAME --build``) writes a
  manifest + a generated registration source and, for ``.hip`` kernel sources, code objects
  (``hipcc --genco``) loadable through the AOT runtime (csrc/runtime/aot_runtime_hip.cc).
"""
from __future__ import annotations

import argparse
import ctypes
import importlib
import importlib.util
import json
import re
import shutil
import subprocess
import sys
import types
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional

from .. import _lib

_SCALARS = {"i1", "i8", "i16", "i32", "i64", "u8", "u16", "u32", "u64", "fp16", "bf16", "fp32", "fp64"}
_POINTEES = _SCALARS | {"fp8e4nv", "fp8e5", "fp8e4b15", "void"}
_SPECS: Dict[str, dict] = {}
_LAUNCHERS: Dict[str, Callable] = {}


class SpecError(ValueError):
    pass


def parse_signature(sig: str) -> List[dict]:
    """``"*fp16:16, i32:1, fp32, %BLOCK"`` -> [{kind: ptr|scalar|const, dtype, hint, name}]."""
    out = []
    for raw in (s.strip() for s in sig.split(",")):
        if not raw:
            continue
        if raw.startswith("%"):
            if not re.fullmatch(r"%[A-Za-z_]\w*", raw):
                raise SpecError(f"bad constant {raw!r}")
            out.append({"kind": "const", "name": raw[1:]})
            continue
        body, _, hint = raw.partition(":")
        if hint and hint not in ("1", "16"):
            raise SpecError(f"bad specialisation hint in {raw!r} (only :1 and :16)")
        if body.startswith("*"):
            if body[1:] not in _POINTEES:
                raise SpecError(f"unknown pointee type in {raw!r}")
            out.append({"kind": "ptr", "dtype": body[1:], "hint": int(hint) if hint else 0})
        else:
            if body not in _SCALARS:
                raise SpecError(f"unknown scalar type in {raw!r}")
            out.append({"kind": "scalar", "dtype": body, "hint": int(hint) if hint else 0})
    return out


def validate_spec(name: str, spec: dict) -> None:
    if not re.fullmatch(r"[A-Za-z_]\w*", name):
        raise SpecError(f"kernel name {name!r} is not a C identifier")
    for key in ("signature", "grid", "triton_algo_infos"):
        if key not in spec:
            raise SpecError(f"{name}: missing {key!r}")
    args = parse_signature(spec["signature"])
    if len(spec["grid"]) != 3:
        raise SpecError(f"{name}: grid needs three expressions")
    consts = {a["name"] for a in args if a["kind"] == "const"}
    infos = spec["triton_algo_infos"]
    if not infos:
        raise SpecError(f"{name}: at least one algo-info is required")
    for info in infos:
        missing = consts - set(info)
        if missing:
            raise SpecError(f"{name}: algo-info lacks constants {sorted(missing)}")
    for expr in spec["grid"]:
        for ref in re.findall(r"%([A-Za-z_]\w*)", str(expr)):
            if ref not in consts and not all(ref in info for info in infos):
                raise SpecError(f"{name}: grid uses unknown constant %{ref}")


def _consts_str(info: dict) -> str:
    return ";".join(f"{k}={int(v)}" for k, v in info.items() if isinstance(v, (int, bool)) or float(v).is_integer())


def eval_grid(spec: dict, info: dict, runtime_args: Dict[str, Any]) -> tuple:
    """Evaluate the three grid expressions (C syntax subset: + - * / % < > ?: parentheses)."""
    env = {**{k: v for k, v in runtime_args.items()}, **info}

    def one(expr: str) -> int:
        e = re.sub(r"%([A-Za-z_]\w*)", r"\1", str(expr))
        # C ternary a ? b : c  ->  (b if a else c); handles the non-nested uses of the specs
        while "?" in e:
            m = re.search(r"\(([^()?]*)\?([^():]*):([^()]*)\)", e)
            if not m:
                m = re.fullmatch(r"([^?]*)\?([^:]*):(.*)", e)
                if not m:
                    raise SpecError(f"cannot parse grid expression {expr!r}")
                e = f"(({m.group(2)}) if ({m.group(1)}) else ({m.group(3)}))"
                break
            e = e[: m.start()] + f"(({m.group(2)}) if ({m.group(1)}) else ({m.group(3)}))" + e[m.end():]
        e = re.sub(r"(?<![/])/(?![/])", "//", e)
        return int(eval(e, {"__builtins__": {}}, env))  # noqa: S307 - expressions come from in-tree specs

    return tuple(one(x) for x in spec["grid"])


# ---- Python-visible namespace ------------------------------------------------------------------
class _AlgoInfoBase:
    """Settable struct of the constants of one kernel (``<kernel>__triton_algo_info_t``)."""

    _fields: tuple = ()

    def __init__(self, **kw):
        for f in self._fields:
            object.__setattr__(self, f, kw.get(f, 0))

    def __setattr__(self, k, v):
        if k not in self._fields:
            raise AttributeError(f"{type(self).__name__} has no field {k!r}")
        object.__setattr__(self, k, v)

    def as_dict(self) -> dict:
        return {f: getattr(self, f) for f in self._fields}


class _Distributed(types.ModuleType):
    """``triton._C.libtriton_distributed.distributed`` stand-in: kernels appear as attributes."""


distributed = _Distributed("libtriton_distributed.distributed")


def _register_native(name: str, spec: dict) -> None:
    lib = _lib.host()
    g = [str(x) for x in spec["grid"]]
    lib.tdh_registry_add_kernel(name.encode(), spec["signature"].encode(), g[0].encode(), g[1].encode(), g[2].encode())
    for info in spec["triton_algo_infos"]:
        lib.tdh_registry_add_algo(name.encode(), _consts_str(info).encode(), name.encode())


def select_algo(name: str, want: dict) -> int:
    return int(_lib.host().tdh_registry_select_algo(name.encode(), _consts_str(want).encode()))


def register_kernel(name: str, spec: dict, launcher: Optional[Callable] = None) -> None:
    validate_spec(name, spec)
    _SPECS[name] = spec
    _register_native(name, spec)
    fields = tuple(dict.fromkeys(k for info in spec["triton_algo_infos"] for k in info))
    info_cls = type(f"{name}__triton_algo_info_t", (_AlgoInfoBase,), {"_fields": fields})
    setattr(distributed, f"{name}__triton_algo_info_t", info_cls)
    if launcher is not None:
        _LAUNCHERS[name] = launcher
    nargs = sum(1 for a in parse_signature(spec["signature"]) if a["kind"] != "const")

    def call(stream, *args):
        if len(args) != nargs + 1:
            raise TypeError(f"{name} expects stream, {nargs} arguments and an algo_info (got {len(args)} after stream)")
        *rt, algo = args
        info = algo.as_dict() if isinstance(algo, _AlgoInfoBase) else dict(algo)
        if select_algo(name, info) < 0:
            raise RuntimeError(f"{name}: no compiled variant matches algo_info {info}; registered: "
                               f"{spec['triton_algo_infos']}")
        fn = _LAUNCHERS.get(name)
        if fn is None:
            raise RuntimeError(f"{name} is registered but has no launcher bound (bind_launcher)")
        return fn(stream, *rt, info)

    call.__name__ = name
    setattr(distributed, name, call)


def bind_launcher(name: str, fn: Callable) -> None:
    _LAUNCHERS[name] = fn


def aot_compile_spaces(spaces: Dict[str, dict]):
    """Decorator of the reference (tools/compile_aot.py:61-77): attaches + registers the specs.
    The decorated callable, when given, is the launcher ``fn(stream, *runtime_args, algo_info_dict)``
    shared by all names in ``spaces`` (it receives ``c_kernel_name=...`` when it accepts it)."""
    for name, spec in spaces.items():
        validate_spec(name, spec)

    def deco(fn):
        fn.__aot_compile_spaces__ = spaces
        for name, spec in spaces.items():
            register_kernel(name, spec, (lambda n: lambda stream, *a: fn(stream, *a, c_kernel_name=n))(name))
        return fn

    return deco


def registered_kernels() -> List[str]:
    lib = _lib.host()
    lib.tdh_registry_kernel_name.restype = ctypes.c_char_p
    return [lib.tdh_registry_kernel_name(i).decode() for i in range(lib.tdh_registry_num_kernels())]


def install_triton_alias() -> None:
    """Expose ``triton._C.libtriton_distributed.distributed`` (the reference's import path) when no
    real ``triton`` package is installed."""
    if "triton" in sys.modules and not getattr(sys.modules["triton"], "__tdist_alias__", False):
        return
    t, c, l = types.ModuleType("triton"), types.ModuleType("triton._C"), types.ModuleType("triton._C.libtriton_distributed")
    t.__tdist_alias__ = True
    t._C, c.libtriton_distributed, l.distributed = c, l, distributed
    sys.modules.update({"triton": t, "triton._C": c, "triton._C.libtriton_distributed": l,
                        "triton._C.libtriton_distributed.distributed": distributed})


# ---- CLI ------------------------------------------------------------------------------------------
_GEN_TEMPLATE = """// generated by triton_dist.tools.compile_aot - registers AOT kernel specs at load time
#include <cstdint>
extern "C" int tdh_registry_add_kernel(const char*, const char*, const char*, const char*, const char*);
extern "C" int tdh_registry_add_algo(const char*, const char*, const char*);
namespace {{
struct Registrar {{
  Registrar() {{
{body}
  }}
}} g_registrar;
}}  // namespace
"""


def _import_file(path: str):
    """Import ``path`` as a module; files that live inside a package are imported under their dotted
    name so that their relative imports work."""
    f = Path(path).resolve()
    parts, d = [f.stem], f.parent
    while (d / "__init__.py").exists():
        parts.insert(0, d.name)
        d = d.parent
    if len(parts) > 1:
        if str(d) not in sys.path:
            sys.path.insert(0, str(d))
        return importlib.import_module(".".join(parts))
    spec = importlib.util.spec_from_file_location(f.stem + "_aot", str(f))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _load_py_specs(path: str, kernel: str) -> Dict[str, dict]:
    fn = getattr(_import_file(path), kernel)
    spaces = getattr(fn, "__aot_compile_spaces__", None)
    if spaces is None:
        raise SpecError(f"{path}:{kernel} is not decorated with aot_compile_spaces")
    return spaces


def _cq(s: str) -> str:
    return '"' + s.replace("\\", "\\\\").replace('"', '\\"') + '"'


def make_kernel_algo_info_struct_name(kernel_name: str) -> str:
    """Name of the settable algo-info struct of a kernel (reference: tools/compile_aot.py:187)."""
    return f"{kernel_name}__triton_algo_info_t"


_C_TYPES = {bool: "bool", int: "int", float: "float", str: "const char*"}


def make_algo_info_decl(kernel_name: str, algo_info: dict) -> str:
    """C declaration of the algo-info struct, one member per constant, for C / C++ callers of the generated
    library (reference: tools/compile_aot.py:190-201)."""
    members = "\n".join(f"  {_C_TYPES.get(type(v), 'int')} {k};" for k, v in algo_info.items())
    return f"struct {make_kernel_algo_info_struct_name(kernel_name)} {{\n{members}\n}};\n"


def aot_runtime_path() -> Path:
    """Source of the code-object loader that generated libraries link against."""
    return Path(__file__).resolve().parent.parent / "csrc" / "runtime" / "aot_runtime_hip.cc"


def libhip_dirs() -> List[str]:
    """Directories holding libamdhip64.so."""
    cands = [Path("/opt/rocm/lib"), Path("/opt/rocm/lib64")]
    return [str(d) for d in cands if (d / "libamdhip64.so").exists()]


libcuda_dirs = libhip_dirs          # the reference's name (tools/compile_aot.py:681)


def compile_kernel(path: str, kernel: str, workspace: Path, arch: str = "gfx950", build: bool = True) -> dict:
    """One ``.hip`` kernel source -> code object ``<workspace>/<stem>.hsaco``; returns its manifest entry."""
    co = Path(workspace) / f"{Path(path).stem}.hsaco"
    if build:
        cmd = ["hipcc", "--genco", f"--offload-arch={arch}", "-O3", "-std=c++17",
               f"-I{Path(__file__).resolve().parent.parent / 'csrc' / 'include'}", path, "-o", str(co)]
        subprocess.run(cmd, check=True)
    return {"source": path, "kernel": kernel, "hsaco": str(co)}


def gen_kernel_library(workspace: Path, library: str, specs: Dict[str, dict]) -> Path:
    """Write ``<library>_registry.cc``: a static registrar that adds every spec (signature, grid, algo-infos) to
    the native registry when the generated library is loaded."""
    body = []
    for name, spec in specs.items():
        g = [str(x) for x in spec["grid"]]
        body.append(f"    tdh_registry_add_kernel({_cq(name)}, {_cq(spec['signature'])}, {_cq(g[0])}, {_cq(g[1])}, "
                    f"{_cq(g[2])});")
        for info in spec["triton_algo_infos"]:
            body.append(f"    tdh_registry_add_algo({_cq(name)}, {_cq(_consts_str(info))}, {_cq(name)});")
    gen = Path(workspace) / f"{library}_registry.cc"
    gen.write_text(_GEN_TEMPLATE.format(body="\n".join(body)))
    header = Path(workspace) / f"{library}_algo_infos.h"
    header.write_text("#pragma once\n" + "".join(make_algo_info_decl(n, sp["triton_algo_infos"][0])
                                                  for n, sp in specs.items() if sp["triton_algo_infos"]))
    return gen


def gen_cmakelists(workspace: Path, library: str) -> Path:
    """CMake project that builds the generated registrar into ``lib<library>.so`` against libtdist_host.so."""
    from .. import build as b

    txt = f"""cmake_minimum_required(VERSION 3.18)
project({library} LANGUAGES CXX)
set(CMAKE_CXX_STANDARD 17)
add_library({library} SHARED {library}_registry.cc)
target_link_directories({library} PRIVATE {b.HOST_LIB.parent})
target_link_libraries({library} PRIVATE :{b.HOST_LIB.name})
set_target_properties({library} PROPERTIES BUILD_RPATH {b.HOST_LIB.parent} INSTALL_RPATH {b.HOST_LIB.parent})
"""
    out = Path(workspace) / "CMakeLists.txt"
    out.write_text(txt)
    return out


def link_all(workspace: Path, library: str) -> Path:
    """Compile and link the generated registrar: ``<workspace>/lib<library>.so``."""
    from .. import build as b

    b.build_host(verbose=False)
    out = Path(workspace) / f"lib{library}.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", str(Path(workspace) / f"{library}_registry.cc"), "-o",
                    str(out), f"-L{b.HOST_LIB.parent}", f"-l:{b.HOST_LIB.name}", f"-Wl,-rpath,{b.HOST_LIB.parent}"], check=True)
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="compile_aot", description=__doc__.split("\n")[0])
    ap.add_argument("--workspace", required=True)
    ap.add_argument("--from-scratch", action="store_true")
    ap.add_argument("--kernels", nargs="+", required=True, help="path.py:kernel (spec) or path.hip:kernel (source)")
    ap.add_argument("--library", default="triton_distributed_kernel")
    ap.add_argument("--patch-cuda-stub", action="store_true", help="accepted for CLI compatibility; no-op")
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--arch", default="gfx950")
    args = ap.parse_args(argv)
    ws = Path(args.workspace)
    if args.from_scratch and ws.exists():
        shutil.rmtree(ws)
    ws.mkdir(parents=True, exist_ok=True)
    manifest = {"library": args.library, "arch": args.arch, "kernels": {}, "code_objects": []}
    specs: Dict[str, dict] = {}
    for item in args.kernels:
        path, _, kernel = item.rpartition(":")
        if not path:
            raise SystemExit(f"--kernels entries are path:kernel, got {item!r}")
        if path.endswith((".hip", ".cpp", ".cu")):
            manifest["code_objects"].append(compile_kernel(path, kernel, ws, args.arch, args.build))
            continue
        for name, spec in _load_py_specs(path, kernel).items():
            manifest["kernels"][name] = {"signature": spec["signature"], "grid": [str(g) for g in spec["grid"]],
                                         "algo_infos": spec["triton_algo_infos"], "source": f"{path}:{kernel}"}
            specs[name] = spec
    (ws / "manifest.json").write_text(json.dumps(manifest, indent=1, default=str))
    gen_kernel_library(ws, args.library, specs)
    gen_cmakelists(ws, args.library)
    if args.build:
        print(f"built {link_all(ws, args.library)}")
    print(f"wrote {ws / 'manifest.json'} ({len(manifest['kernels'])} kernels, {len(manifest['code_objects'])} code objects)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
