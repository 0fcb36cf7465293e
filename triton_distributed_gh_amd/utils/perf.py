"""Timing, data generation, rank-aware printing and comparison helpers.

Same helpers the reference's tests are written against (reference: python/triton_dist/utils.py -
``perf_func`` :274-286, ``dist_print`` :289-318, ``generate_data`` :222-259, ``assert_allclose``
:870-899, ``sleep_async`` :1018-1020), working on both the GPU and the CPU backends.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def perf_func(func: Callable, iters: int, warmup_iters: int):
    """Return (last_output, mean milliseconds per call) using device events on GPU, wall clock on CPU."""
    out = None
    if torch.cuda.is_available():
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for n in range(iters + warmup_iters):
            if n == warmup_iters:
                start.record()
            out = func()
        stop.record()
        stop.synchronize()
        return out, start.elapsed_time(stop) / max(iters, 1)
    for _ in range(warmup_iters):
        out = func()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = func()
    return out, (time.perf_counter() - t0) * 1e3 / max(iters, 1)


def graph_bench(fn: Callable, n: int = 20, reps: int = 10) -> float:
    """Microseconds per call of ``fn`` with ``n`` calls captured in one HIP graph: kernel time without the host's
    launch cost (``fn`` may rotate over input copies through a counter of its own to keep its operands cold)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3


def sleep_async(duration_ms: float) -> None:
    """Queue a busy kernel on the current stream so the host can run ahead of the device."""
    if torch.cuda.is_available():
        # torch.cuda._sleep counts device clock ticks of the 100 MHz wall clock on ROCm builds
        torch.cuda._sleep(int(duration_ms * 1e5))


def dist_print(*args, allowed_ranks: Iterable[int] | str = (0,), prefix: bool = False, need_sync: bool = False,
               **kwargs) -> None:
    """Print from selected ranks, optionally in rank order (one barrier per rank when need_sync)."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if allowed_ranks == "all":
        allowed = list(range(world))
    else:
        allowed = list(allowed_ranks)
    head = (f"[rank:{rank}]",) if prefix else ()
    if not (need_sync and dist.is_initialized()):
        if rank in allowed:
            print(*head, *args, **kwargs, flush=True)
        return
    for r in range(world):
        dist.barrier()
        if r == rank and rank in allowed:
            print(*head, *args, **kwargs, flush=True)


def _make_tensor(shape: Sequence[int], dtype: torch.dtype, init_args, device) -> torch.Tensor:
    """init_args: (scale, bias) -> uniform[-1,1)*scale + bias; for integer dtypes (low, high)."""
    if init_args is None:
        return torch.empty(shape, dtype=dtype, device=device)
    if dtype in (torch.int8, torch.int16, torch.int32, torch.int64, torch.uint8):
        lo, hi = init_args
        return torch.randint(lo, hi, shape, dtype=dtype, device=device)
    scale, bias = init_args
    t = (torch.rand(shape, dtype=torch.float32, device=device) * 2 - 1) * scale + bias
    return t.to(dtype)


def generate_data(configs, device=None):
    """Endless generator of tuples of tensors; configs = [(shape, dtype, init_args), ...]."""
    device = device or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu")
    while True:
        yield tuple(_make_tensor(shape, dtype, init, device) for shape, dtype, init in configs)


def assert_allclose(x: torch.Tensor, y: torch.Tensor, rtol: float = 1e-2, atol: float = 1e-2, verbose: bool = True):
    """torch.allclose with a readable report of the worst elements on failure."""
    if x.shape != y.shape:
        raise AssertionError(f"shape mismatch {tuple(x.shape)} vs {tuple(y.shape)}")
    xf, yf = x.float(), y.float()
    ok = torch.isclose(xf, yf, rtol=rtol, atol=atol)
    if bool(ok.all()):
        return
    bad = (~ok).nonzero()
    msg = [f"{bad.shape[0]} / {x.numel()} elements differ (rtol {rtol}, atol {atol})"]
    if verbose:
        diff = (xf - yf).abs()
        worst = torch.topk(diff.flatten(), k=min(5, diff.numel())).indices
        for idx in worst.tolist():
            pos = tuple(int(v) for v in torch.unravel_index(torch.tensor(idx), x.shape))
            msg.append(f"  at {pos}: {xf.flatten()[idx].item():.6f} vs {yf.flatten()[idx].item():.6f}")
    raise AssertionError("\n".join(msg))


def bitwise_equal(x: torch.Tensor, y: torch.Tensor) -> bool:
    return x.shape == y.shape and x.dtype == y.dtype and bool(torch.equal(x, y))
