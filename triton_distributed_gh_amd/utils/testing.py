"""Test harness: run every rank of a multi-rank op inside ONE process.

``VirtualCluster(w)`` builds ``w`` virtual ranks (see ``create_virtual_node``).  ``run(fn)`` calls
``fn(rank, env)`` once per rank:

* hip backend  - sequentially on the host, each rank under its own stream; launches are
  asynchronous, so all ranks' kernels are in flight together and meet through device-side flags;
* host backend - one Python thread per rank (the CPU protocol blocks in C with the GIL released).

A rank that spins on a peer must not sit in front of that peer's work in the same hardware queue.
torch's stream pool (64 streams per device) is mapped by HIP onto GPU_MAX_HW_QUEUES queues, so pool
streams can collide; the harness therefore creates its own 2 streams per rank (compute + side)
ONCE per process, straight from HIP, and hands them out as ``torch.cuda.ExternalStream``.

This complements the torchrun-launched multi-process scripts (which exercise hipIpc and gloo/RCCL
bootstrap): it needs no second process and no IPC, so it runs under plain ``pytest`` on a 1-GPU box.
The reference has no such harness - all its tests need one real GPU per rank (reference:
python/triton_dist/test/amd/*.py launched by scripts/launch_amd.sh).
"""
from __future__ import annotations

import ctypes
import sys
import threading
from typing import Callable, List, Optional

import torch

from .. import _lib
from .distributed import create_virtual_node, use_env

_POOL = {"main": [], "side": [], "gate": []}


def _pool_streams(kind: str, n: int):
    lib = _lib.hip()
    while len(_POOL[kind]) < n:
        h = ctypes.c_void_p()
        _lib.check(lib.tdist_stream_create(ctypes.byref(h), 0), "hipStreamCreate")
        _POOL[kind].append(torch.cuda.ExternalStream(h.value))
    return _POOL[kind][:n]


class VirtualCluster:
    def __init__(self, world_size: int, heap_bytes: int = 1 << 28, sig_bytes: int = 1 << 22, backend=None):
        self.envs = create_virtual_node(world_size, heap_bytes, sig_bytes, backend)
        self.world_size = world_size
        self.backend = self.envs[0].backend
        if self.backend == "hip":
            # interleave creation so rank r's two streams get neighbouring hardware queues
            for r in range(world_size):
                _pool_streams("main", r + 1)
                _pool_streams("side", r + 1)
            self.streams = _pool_streams("main", world_size)
            self.side_streams = _pool_streams("side", world_size)
            for r, env in enumerate(self.envs):
                env.side_stream = self.side_streams[r]
        else:
            self.streams = self.side_streams = None

    def run(self, fn: Callable, sync: bool = True) -> List:
        outs = [None] * self.world_size
        if self.backend == "hip":
            # The rank streams do not synchronise with the caller's stream (and must not wait for an event of it: see
            # timed_run), so whatever the caller queued there - the fill of a fresh context's counters, the kernels
            # that produce the inputs - has to be complete before a rank's kernel may read it.  Without this wait a
            # fused kernel occasionally started while the bf16 conversion of its freshly drawn input was still
            # running: one wrong result in several thousand test runs.
            torch.cuda.current_stream().synchronize()
            for r, env in enumerate(self.envs):
                with torch.cuda.stream(self.streams[r]), use_env(env):
                    outs[r] = fn(r, env)
            if sync:
                torch.cuda.synchronize()
            return outs
        errs = [None] * self.world_size

        def work(r):
            try:
                with use_env(self.envs[r]):
                    outs[r] = fn(r, self.envs[r])
            except BaseException as e:  # noqa: BLE001 - re-raised below on the main thread
                errs[r] = e

        threads = [threading.Thread(target=work, args=(r,)) for r in range(self.world_size)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in errs:
            if e is not None:
                raise e
        return outs

    def timed_run(self, fn: Callable, gate_cycles: int = 5_000_000) -> float:
        """One call of ``fn(rank, env)`` per rank, all released at the same instant; returns the
        milliseconds from the release until every rank has finished.

        The host enqueues the ranks one after the other, so without a gate rank 0 would start
        early and the time would include the launch skew of the others.  Here every rank's stream
        first waits for one event that a gate stream records after a ~2 ms sleep kernel; by then
        all ranks' work is queued.  The clock is the host's (it wakes on the gate event, then on
        the device going idle), good to ~10 us; per-kernel times come from ``rocprofv3
        --kernel-trace``.  No event is recorded on the rank streams: an event is a barrier packet,
        and when virtual ranks share a hardware queue it would hold a peer's kernel behind the
        kernel that waits for it.  Exactly one call per rank is in flight for the same reason."""
        assert self.backend == "hip"
        import time

        torch.cuda.current_stream().synchronize()        # (the inputs: see run())
        gate = _pool_streams("gate", 1)[0]
        ev_gate = torch.cuda.Event()
        with torch.cuda.stream(gate):
            torch.cuda._sleep(gate_cycles)
            ev_gate.record()
        for r, env in enumerate(self.envs):
            with torch.cuda.stream(self.streams[r]):
                self.streams[r].wait_event(ev_gate)
                fn(r, env)
        ev_gate.synchronize()
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    def collective(self, fn: Callable) -> List:
        """Run a collective constructor (symmetric allocation) rank by rank on the host."""
        return [fn(r, env) for r, env in enumerate(self.envs)]

    def close(self) -> None:
        if sys.exc_info()[0] is not None:
            # A test is failing: its report may still touch symmetric tensors (reading freed device
            # memory is a GPU fault), so leak the arenas instead of freeing them.
            return
        if self.backend == "hip":
            try:
                torch.cuda.synchronize()
            except Exception:
                return  # the device is in an error state: freeing arenas now could fault again
        for env in self.envs:
            env.heap.close()


def assert_close_elementwise(got: torch.Tensor, ref: torch.Tensor, rtol: float = 1e-2, atol: float = 1e-2,
                             what: str = "", magnitude: Optional[torch.Tensor] = None) -> None:
    """Every element within ``atol + rtol * |ref|`` (the reference's ``assert_allclose(atol=rtol=1e-2)``
    convention, e.g. python/triton_dist/test/amd/test_ag_gemm_intra_node.py:170-190).  A metric relative to
    the largest reference element would let a wrong small-magnitude row through; this one does not.  The
    message names the worst element.  ``magnitude`` (same shape) replaces ``|ref|`` in the bound where the
    result is a sum of separately rounded 16-bit partials: their rounding errors scale with the sum of the
    partials' magnitudes, not with the (possibly cancelling) total."""
    g, r = got.float(), ref.float()
    assert g.shape == r.shape, f"{what}: shape {tuple(g.shape)} vs {tuple(r.shape)}"
    excess = (g - r).abs() - (atol + rtol * (r.abs() if magnitude is None else magnitude.float()))
    bad = excess > 0
    if bool(bad.any()):
        flat = int(excess.argmax())
        idx = tuple(int(i) for i in torch.unravel_index(torch.tensor(flat), g.shape))
        raise AssertionError(f"{what}: {int(bad.sum())} of {g.numel()} elements outside atol={atol} rtol={rtol}; worst at "
                             f"{idx}: got {g[idx].item():.6g}, expected {r[idx].item():.6g}")
