"""Harness shared by the command-line test / benchmark scripts (triton_dist/test/**).

A script describes per-rank work as ``fn(rank, env)``.  ``Runner`` executes it either
* under ``torchrun`` - one process per GPU, RCCL + gloo + hipIpc symmetric heap (the way the
  reference launches every test: scripts/launch_amd.sh), or
* with ``--virtual-ranks N`` - N ranks inside this one process on one GPU (or on the CPU backend),
  which is how multi-rank protocols are exercised on a single-GPU machine.
"""
from __future__ import annotations

import argparse
import os
from typing import Callable, List

import torch
import torch.distributed as dist

from .distributed import finalize_distributed, initialize_distributed
from .perf import dist_print  # noqa: F401
from .testing import VirtualCluster


def add_common_args(ap: argparse.ArgumentParser) -> argparse.ArgumentParser:
    ap.add_argument("--virtual-ranks", type=int, default=0, help="run N ranks inside this process (no torchrun)")
    ap.add_argument("--heap-mb", type=int, default=2048, help="symmetric heap per rank")
    ap.add_argument("--warmup", default=5, type=int)
    ap.add_argument("--iters", default=10, type=int)
    ap.add_argument("--check", action="store_true", help="correctness run against the torch / RCCL golden")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--seed", type=int, default=42)
    return ap


MAX_HW_QUEUES = 32                                   # upper bound this library ever asks HIP for
MAX_VIRTUAL_RANKS = (MAX_HW_QUEUES - 4) // 2          # 14


class Runner:

    def __init__(self, args):
        self.args = args
        self.vc = None
        self._harness_ms = None
        if args.virtual_ranks and args.virtual_ranks > 0 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
            # Every virtual rank's two streams need hardware queues of their own: kernels that
            # share a queue start in order with limited overlap, so a rank spinning on a peer
            # queued behind it in the same hardware queue would wait for its spin budget.
            # A smaller value in the environment (some machines export HIP's default of 4) is raised,
            # never past MAX_HW_QUEUES: more ranks than that allows are refused instead.
            if args.virtual_ranks > MAX_VIRTUAL_RANKS:
                raise SystemExit(f"--virtual-ranks {args.virtual_ranks}: at most {MAX_VIRTUAL_RANKS} virtual ranks fit "
                                 f"{MAX_HW_QUEUES} hardware queues (two streams per rank plus four for the harness)")
            need = min(MAX_HW_QUEUES, max(8, 2 * args.virtual_ranks + 4))
            if int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < need:
                os.environ["GPU_MAX_HW_QUEUES"] = str(need)
            self.vc = VirtualCluster(args.virtual_ranks, heap_bytes=args.heap_mb << 20)
            self.world_size, self.envs = args.virtual_ranks, self.vc.envs
            self.device = self.envs[0].device
            torch.manual_seed(args.seed)
        else:
            env = initialize_distributed(seed=args.seed, heap_bytes=args.heap_mb << 20)
            self.world_size, self.envs, self.device = env.world_size, [env], env.device

    @property
    def virtual(self) -> bool:
        return self.vc is not None

    @property
    def local_ranks(self) -> List[int]:
        return list(range(self.world_size)) if self.virtual else [self.envs[0].rank]

    @property
    def is_main(self) -> bool:
        return self.virtual or self.envs[0].rank == 0

    def collective(self, fn: Callable) -> dict:
        """{rank: fn(rank, env)} for the ranks hosted by this process (context creation)."""
        if self.virtual:
            return dict(enumerate(self.vc.collective(fn)))
        env = self.envs[0]
        return {env.rank: fn(env.rank, env)}

    def run(self, fn: Callable) -> dict:
        if self.virtual:
            return dict(enumerate(self.vc.run(fn)))
        env = self.envs[0]
        out = fn(env.rank, env)
        if env.backend == "hip":
            torch.cuda.synchronize()
        return {env.rank: out}

    def all_gather(self, per_rank: dict) -> List[torch.Tensor]:
        """Every rank's tensor, in rank order, on this process (golden construction)."""
        if self.virtual:
            return [per_rank[r] for r in range(self.world_size)]
        env = self.envs[0]
        t = per_rank[env.rank]
        if self.world_size == 1:
            return [t]
        outs = [torch.empty_like(t.cpu()) for _ in range(self.world_size)]
        dist.all_gather(outs, t.cpu().contiguous(), group=env.cpu_group)
        return [o.to(t.device) for o in outs]

    def time(self, fn: Callable, warmup: int, iters: int) -> float:
        """Mean milliseconds per call of ``fn(rank, env)`` over all hosted ranks together."""
        for _ in range(warmup):
            self.run(fn)
        self.barrier()
        import time as _t

        cuda = self.device.type == "cuda"
        if self.virtual and cuda:
            # gate-released, one call per rank in flight; median over the iterations (host clock),
            # net of what the same harness measures for an empty call (event wake-up + device sync)
            if self._harness_ms is None:
                base = sorted(self.vc.timed_run(lambda r, env: None) for _ in range(15))
                self._harness_ms = base[len(base) // 2]
            per_iter = sorted(self.vc.timed_run(fn) for _ in range(iters))
            return max(per_iter[len(per_iter) // 2] - self._harness_ms, 1e-4)
        if self.virtual:
            t0 = _t.perf_counter()
            for _ in range(iters):
                self.vc.run(fn)
            return (_t.perf_counter() - t0) * 1e3 / iters
        env = self.envs[0]
        if cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn(env.rank, env)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters
        t0 = _t.perf_counter()
        for _ in range(iters):
            fn(env.rank, env)
        return (_t.perf_counter() - t0) * 1e3 / iters

    def barrier(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        if not self.virtual and self.world_size > 1:
            dist.barrier(group=self.envs[0].cpu_group)

    def print(self, *a):
        if self.is_main:
            print(*a, flush=True)

    def print_all(self, *a):
        prefix = "" if self.virtual else f"[rank {self.envs[0].rank}] "
        print(prefix + " ".join(str(x) for x in a), flush=True)

    def close(self):
        if self.virtual:
            self.vc.close()
        else:
            finalize_distributed()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return (a.float() - b.float()).abs().max().item() / max(b.float().abs().max().item(), 1e-9)
