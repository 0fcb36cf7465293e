"""List the public host-level names of the reference's python modules that the ``triton_dist`` facade lacks.

A name counts when it is a top-level ``def`` / ``class`` of a reference module, does not start with ``_`` and
is not a device function (decorated with a ``*.jit`` decorator - device code has no counterpart by NAME here,
it lives in the HIP sources).  Usage: ``python scripts/dev/facade_gap.py [/root/reference/python/triton_dist]``.
"""
import ast
import importlib
import sys
from pathlib import Path

SKIP_DIRS = {"test", "benchmark", "tutorials"}


def host_names(path: Path):
    tree = ast.parse(path.read_text())
    out = []
    for node in tree.body:
        if not isinstance(node, (ast.FunctionDef, ast.ClassDef)) or node.name.startswith("_"):
            continue
        decos = [ast.unparse(d) for d in node.decorator_list]
        if any("jit" in d or "core.extern" in d or "builtin" in d for d in decos):
            continue
        out.append(node.name)
    return out


def gaps(root: Path):
    missing = {}
    for f in sorted(root.rglob("*.py")):
        rel = f.relative_to(root)
        if SKIP_DIRS & set(rel.parts):
            continue
        names = host_names(f)
        if not names:
            continue
        mod = ".".join(("triton_dist",) + rel.with_suffix("").parts)
        if mod.endswith(".__init__"):
            mod = mod[: -len(".__init__")]
        try:
            m = importlib.import_module(mod)
        except ImportError:
            missing[mod] = ["<module>"] + names
            continue
        lack = [n for n in names if not hasattr(m, n)]
        if lack:
            missing[mod] = lack
    return missing


if __name__ == "__main__":
    root = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/python/triton_dist")
    sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
    total = 0
    for mod, names in gaps(root).items():
        total += len(names)
        print(f"{mod}: {' '.join(names)}")
    print(f"-- {total} names missing")
