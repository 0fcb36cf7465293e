"""Profiling helpers: torch profiler contexts, merged multi-rank chrome traces, rocprofv3 recipes.

``group_profile`` mirrors the reference's context manager (reference: python/triton_dist/utils.py
:505-589; merge helpers :342-502): every rank records a chrome trace, the traces are gathered on
rank 0, pids / tids are re-labelled per rank and one ``prof/<name>_merged.json.tar.gz`` is written.
``ROCPROF_RECIPES`` lists the counter sets used for the kernels in this repository (each fits the
per-block PMC slot limits of gfx950 so it is ONE rocprofv3 pass).
"""
from __future__ import annotations

import contextlib
import gzip
import io
import json
import os
import tarfile
from typing import Optional

import torch
import torch.distributed as dist

ROCPROF_RECIPES = {
    # per-kernel time
    "time": "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py",
    # MFMA utilisation and stall buckets (8 SQ counters)
    "mfma": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 "
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv "
            "-d gpurun_out/pmc_mfma -- python3 bench.py --steps 3 --warmup 1",
    # LDS bank conflicts
    "lds": "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS "
           "--output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --steps 3 --warmup 1",
    # HBM / fabric bytes (FETCH_SIZE uses 3 of the 4 TCC slots; WRITE_SIZE goes in its own pass)
    "hbm_read": "rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_rd -- python3 bench.py --steps 3 --warmup 1",
    "hbm_write": "rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_wr "
                 "-- python3 bench.py --steps 3 --warmup 1",
}


def get_torch_prof_ctx(do_prof: bool):
    """torch.profiler context (CPU + GPU activities) or a null context."""
    if not do_prof:
        return contextlib.nullcontext()
    acts = [torch.profiler.ProfilerActivity.CPU]
    if torch.cuda.is_available():
        acts.append(torch.profiler.ProfilerActivity.CUDA)
    return torch.profiler.profile(activities=acts, record_shapes=True, with_stack=False)


def _relabel(trace: dict, rank: int) -> list:
    """Shift pids so ranks do not collide and tag process names with the rank."""
    events = trace.get("traceEvents", [])
    out = []
    for ev in events:
        ev = dict(ev)
        pid = ev.get("pid")
        if isinstance(pid, int):
            ev["pid"] = pid + rank * 10_000_000
        elif pid is not None:
            ev["pid"] = f"rank{rank}:{pid}"
        if ev.get("ph") == "M" and ev.get("name") == "process_name":
            args = dict(ev.get("args", {}))
            args["name"] = f"[rank {rank}] {args.get('name', '')}"
            ev["args"] = args
        out.append(ev)
    return out


class ParallelJsonDumper:
    """Write a dict whose ``parallel_field`` is a very long list (the ``traceEvents`` of a merged chrome trace)
    without building the whole JSON text in memory: the list is serialised in chunks by a thread pool and
    streamed to the file in order (reference: python/triton_dist/utils.py ``ParallelJsonDumper``, which
    uses a process pool; ``json.dumps`` of plain dicts releases nothing, but the chunks bound peak memory and the
    file is written while later chunks are still being encoded)."""

    def __init__(self, parallel_field: str, chunk_size: int = 5000, workers: int = 4):
        self.parallel_field, self.chunk_size, self.workers = parallel_field, max(1, chunk_size), max(1, workers)

    def dump(self, data: dict, output_path) -> None:
        from concurrent.futures import ThreadPoolExecutor
        from pathlib import Path

        out = Path(output_path)
        out.parent.mkdir(parents=True, exist_ok=True)
        base = {k: v for k, v in data.items() if k != self.parallel_field}
        items = data.get(self.parallel_field, [])
        chunks = [items[i:i + self.chunk_size] for i in range(0, len(items), self.chunk_size)]
        with ThreadPoolExecutor(self.workers) as ex, open(out, "w") as f:
            head = json.dumps(base, separators=(",", ":"))
            f.write((head[:-1] + ("," if base else "")) + f'"{self.parallel_field}":[')
            first = True
            for text in ex.map(lambda c: json.dumps(c, separators=(",", ":"))[1:-1], chunks):
                if text:
                    f.write(("" if first else ",") + text)
                    first = False
            f.write("]}")


def process_trace_json(trace: dict, rank: int) -> dict:
    """One rank's chrome trace with its process / thread ids re-labelled for the merged view (reference:
    python/triton_dist/utils.py ``process_trace_json``)."""
    out = {k: v for k, v in trace.items() if k != "traceEvents"}
    out["traceEvents"] = _relabel(trace, rank)
    return out


def merge_traces(traces: list) -> dict:
    merged = {"traceEvents": [], "displayTimeUnit": "ms"}
    for rank, t in enumerate(traces):
        if t is None:
            continue
        merged["traceEvents"].extend(_relabel(t, rank))
    return merged


@contextlib.contextmanager
def group_profile(name: Optional[str] = None, do_prof: bool = True, merge_group: bool = True,
                  keep_merged_only: bool = True, compress: bool = True, group=None, out_dir: str = "prof"):
    if not do_prof:
        yield None
        return
    name = name or "trace"
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    prof = get_torch_prof_ctx(True)
    with prof:
        yield prof
    os.makedirs(out_dir, exist_ok=True)
    local_path = os.path.join(out_dir, f"{name}_rank{rank}.json")
    prof.export_chrome_trace(local_path)
    if not merge_group or world == 1:
        if compress:
            with open(local_path, "rb") as f, gzip.open(local_path + ".gz", "wb") as g:
                g.write(f.read())
            os.remove(local_path)
        return
    with open(local_path) as f:
        mine = json.load(f)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0, group=group)
    if keep_merged_only:
        os.remove(local_path)
    if rank == 0:
        merged = merge_traces(gathered)
        data = json.dumps(merged).encode()
        if compress:
            with tarfile.open(os.path.join(out_dir, f"{name}_merged.json.tar.gz"), "w:gz") as tar:
                info = tarfile.TarInfo(f"{name}_merged.json")
                info.size = len(data)
                tar.addfile(info, io.BytesIO(data))
        else:
            with open(os.path.join(out_dir, f"{name}_merged.json"), "wb") as f:
                f.write(data)
