"""Device-ISA inspection: compile one HIP source for gfx950 to assembly (device side only) and cut
out a kernel's body, so tests can check how a primitive was lowered without running it.

Role in the reference: the MLIR lit test of the ``distributed.wait`` lowering
(unittest/lower_wait.mlir:58) and ``test/nvidia/test_distributed_wait.py --case lower`` - there the
check is on the generated LLVM / PTX, here on the gfx950 instructions hipcc produced from the
device functions in csrc/include/tdist/primitives.hpp.
"""
from __future__ import annotations

import functools
import re
import subprocess
from pathlib import Path
from typing import Dict

from .. import build


@functools.lru_cache(maxsize=8)
def device_asm(source: str) -> str:
    """gfx950 assembly of ``csrc/ops/<source>`` built with the library's own flags."""
    src = build.CSRC / "ops" / source
    cmd = [build._hipcc(), *build.HIPCC_FLAGS, "-S", "--cuda-device-only", "-o", "-", str(src)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{res.stderr[-2000:]}")
    return res.stdout


def kernel_bodies(asm: str) -> Dict[str, str]:
    """{demangled-ish kernel name: instruction text} for every ``s_endpgm``-terminated function."""
    out: Dict[str, str] = {}
    cur, lines = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s", line)
        if m and cur is None:
            cur, lines = m.group(1), []
            continue
        if cur is not None:
            code = line.split(";")[0].strip()
            if code and not code.startswith("."):
                lines.append(code)
            if code.startswith("s_endpgm"):
                out[cur] = "\n".join(lines)
                cur = None
    return out


def kernel_isa(source: str, name_fragment: str) -> str:
    """Instruction text of the kernel of ``source`` whose mangled name contains ``name_fragment``."""
    bodies = kernel_bodies(device_asm(source))
    hits = [k for k in bodies if name_fragment in k]
    if len(hits) != 1:
        raise KeyError(f"{name_fragment!r} matches {hits} in {source} (kernels: {sorted(bodies)})")
    return bodies[hits[0]]


def lowering_report() -> Dict[str, bool]:
    """The facts the wait / notify lowering must satisfy (one entry per fact)."""
    wait = kernel_isa("primitives_test.hip", "wait_flag_kernel")
    ring = kernel_isa("primitives_test.hip", "ring_queue_kernel")
    notify = kernel_isa("primitives_test.hip", "notify_kernel")
    return {
        # the poll is a relaxed load that bypasses the non-coherent caches of its scope
        "wait polls with a system-scope load (sc0 sc1)": bool(re.search(r"global_load_dword \S+, \S+, \S+ sc0 sc1", wait)),
        "wait polls with an agent-scope load (sc1) for scope gpu": bool(re.search(r"global_load_dword \S+, \S+, \S+ sc1$", wait, re.M)),
        "wait backs off with s_sleep": "s_sleep" in wait,
        "wait is bounded by the wall clock (s_memrealtime)": "s_memrealtime" in wait,
        # block-collective form: acquire = one cache invalidate after the match, then a barrier
        "acquire after the match is a cache invalidate (buffer_inv sc0 sc1)": "buffer_inv sc0 sc1" in ring,
        "wait ends in a workgroup barrier": "s_barrier" in ring,
        # notify: release (L2 write-back at system scope) before the flag store / add
        "notify releases with buffer_wbl2 sc0 sc1": "buffer_wbl2 sc0 sc1" in notify,
        "notify(set) is a system-scope store": bool(re.search(r"(global|flat)_store_dword .* sc0 sc1", notify)),
        "notify(add) is a system-scope atomic add": bool(re.search(r"(global|flat)_atomic_add\S* .* sc1", notify)),
    }
