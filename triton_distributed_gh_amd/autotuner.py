"""Contextual autotuner: tunes knobs of the AOT kernels by timing the WHOLE calling thunk.

Reference: python/triton_dist/autotuner.py:43-250 and docs/autotuner.md:21-24.  The reference
monkey-patches Triton's per-kernel ``Autotuner.run`` so that each candidate ``triton.Config`` is
timed inside the complete distributed thunk (a fused op cannot be benchmarked kernel by kernel: its
kernels wait on peers), takes the MAX of every timing over the ranks so all ranks pick the same
config, and logs to ``.autotune_logs/rank-N.log``.

There is no JIT here - kernels are built ahead of time with a few runtime knobs (main-loop schedule,
tile-order group size, chunk size, workgroup budget ...).  A knob is read with ``tuning.get(name,
default)`` at the call site; ``contextual_autotune`` sweeps a declared space of knob values around
the decorated thunk with the same rules: whole-thunk timing, cross-rank MAX, one log per rank, the
choice cached per call signature (tensor shapes / dtypes and scalar arguments).
"""
from __future__ import annotations

import functools
import itertools
import os
import time
from pathlib import Path
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence

import torch

_ACTIVE: Dict[str, Any] = {}

# knobs the ops consult; the lists are the default sweep spaces
KNOWN_KNOBS: Dict[str, Sequence[Any]] = {
    "gemm.variant": (3 | (3 << 16), 2, 1),   # persistent two-phase / two phases per K-step / four phases
    "gemm.group_m": (8, 4, 16),        # tile rows per L2 group
    "ag_gemm.M_PER_CHUNK": (256, 512, 1024),
    "ag_gemm.producer": ("kernel", "copy_engine"),
    "ag_gemm.PUSH_WGS": (0, 8, 16, 32),          # 0 = 8 per outgoing link
    "ag_gemm.SPLIT_K": (0, 1, 2, 4),             # 0 = fill the CUs
    "gemm_rs.M_PER_CHUNK": (256, 512),
    "gemm_mid.WN": (0, 1, 2),                    # 0 = planned from the shape
    "gemm_mid.SPLITS": (0, 1, 2, 4, 8),
    "moe_rs.N_CHUNKS": (1, 2, 4),                # column bands of MoE+RS pipelined over two streams
    "moe_rs.RS_CUS": (32, 16, 48),               # compute units the grouped GEMM leaves to the reduce
}


class tuning:
    """Read side: ``tuning.get("gemm.variant", 2)`` returns the value under test / selected, or the
    default outside any tuned thunk."""

    @staticmethod
    def get(name: str, default: Any = None) -> Any:
        return _ACTIVE.get(name, default)

    @staticmethod
    def active() -> Dict[str, Any]:
        return dict(_ACTIVE)


def _signature(args, kwargs) -> tuple:
    def one(x):
        if isinstance(x, torch.Tensor):
            return ("T", tuple(x.shape), str(x.dtype))
        if isinstance(x, (int, float, str, bool, type(None))):
            return x
        return type(x).__name__

    return tuple(one(a) for a in args) + tuple((k, one(v)) for k, v in sorted(kwargs.items()))


class ContextualAutoTuner:

    def __init__(self, fn: Callable, space: Optional[Dict[str, Iterable[Any]]] = None, is_dist: bool = False,
                 n_repeat: int = 5, n_warmup: int = 3, max_configs: int = 32, group=None):
        self.fn = fn
        self.space = {k: list(v) for k, v in (space or {}).items()}
        self.is_dist, self.n_repeat, self.n_warmup, self.max_configs = is_dist, n_repeat, n_warmup, max_configs
        self.group = group
        self.cache: Dict[tuple, Dict[str, Any]] = {}
        self.timings: Dict[tuple, List[float]] = {}
        functools.update_wrapper(self, fn)

    # ---- logging (one file per rank, like the reference) ------------------------------------------
    def _log(self, msg: str) -> None:
        rank = int(os.environ.get("RANK", 0))
        d = Path(os.environ.get("TDIST_AUTOTUNE_LOG_DIR", ".autotune_logs"))
        try:
            d.mkdir(parents=True, exist_ok=True)
            with open(d / f"rank-{rank}.log", "a") as f:
                f.write(msg + "\n")
        except OSError:
            pass

    def configs(self) -> List[Dict[str, Any]]:
        if not self.space:
            return [{}]
        keys = list(self.space)
        out = [dict(zip(keys, vals)) for vals in itertools.product(*(self.space[k] for k in keys))]
        return out[: self.max_configs]

    def _time(self, args, kwargs) -> float:
        cuda = torch.cuda.is_available()
        for _ in range(self.n_warmup):
            self.fn(*args, **kwargs)
        if cuda:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(self.n_repeat):
                self.fn(*args, **kwargs)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / self.n_repeat
        t0 = time.perf_counter()
        for _ in range(self.n_repeat):
            self.fn(*args, **kwargs)
        return (time.perf_counter() - t0) * 1e3 / self.n_repeat

    def _reduce_max(self, times: List[float]) -> List[float]:
        import torch.distributed as dist

        if not (self.is_dist and dist.is_initialized() and dist.get_world_size() > 1):
            return times
        group = self.group
        if group is None:
            try:
                from .utils.distributed import get_env

                group = get_env().cpu_group
            except Exception:
                group = None
        t = torch.tensor(times, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)     # every rank sees the slowest rank's time
        return t.tolist()

    def tune(self, *args, **kwargs) -> Dict[str, Any]:
        key = _signature(args, kwargs)
        if key in self.cache:
            return self.cache[key]
        cfgs = self.configs()
        saved = dict(_ACTIVE)
        times: List[float] = []
        for cfg in cfgs:
            _ACTIVE.clear()
            _ACTIVE.update(saved)
            _ACTIVE.update(cfg)
            try:
                times.append(self._time(args, kwargs))
            except Exception as e:  # noqa: BLE001 - an invalid knob combination is simply not chosen
                self._log(f"[{self.fn.__name__}] config {cfg} failed: {e}")
                times.append(float("inf"))
        _ACTIVE.clear()
        _ACTIVE.update(saved)
        times = self._reduce_max(times)
        best = min(range(len(cfgs)), key=lambda i: times[i])
        self.cache[key], self.timings[key] = cfgs[best], times
        for cfg, t in zip(cfgs, times):
            self._log(f"[{self.fn.__name__}] {key} {cfg} -> {t:.4f} ms")
        self._log(f"[{self.fn.__name__}] {key} best {cfgs[best]} ({times[best]:.4f} ms)")
        return cfgs[best]

    def __call__(self, *args, **kwargs):
        cfg = self.tune(*args, **kwargs)
        saved = dict(_ACTIVE)
        _ACTIVE.update(cfg)
        try:
            return self.fn(*args, **kwargs)
        finally:
            _ACTIVE.clear()
            _ACTIVE.update(saved)


def contextual_autotune(is_dist: bool = False, n_repeat: int = 5, n_warmup: int = 3,
                        space: Optional[Dict[str, Iterable[Any]]] = None, **kw):
    """Decorator: ``@contextual_autotune(is_dist=True, space={"gemm.variant": [1, 2]})``.  Without
    ``space`` the GEMM knobs are swept."""
    if space is None:
        space = {k: KNOWN_KNOBS[k] for k in ("gemm.variant", "gemm.group_m")}

    def deco(fn):
        return ContextualAutoTuner(fn, space=space, is_dist=is_dist, n_repeat=n_repeat, n_warmup=n_warmup, **kw)

    return deco
