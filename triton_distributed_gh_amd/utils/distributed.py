"""Process bootstrap: torch.distributed (RCCL + gloo) and the node-wide symmetric heap.

One call - ``initialize_distributed()`` - does what the reference spreads over
``triton_dist.utils.initialize_distributed`` (reference: python/triton_dist/utils.py:182-205) and the
per-test AMD rocSHMEM bootstrap (python/triton_dist/test/amd/test_gemm_rs_intra_node.py:153-195):
reads RANK / LOCAL_RANK / WORLD_SIZE, creates the process groups, seeds the RNGs and maps the
symmetric heap of every rank.

Backends
* GPU, one rank per device: ``cpu:gloo,cuda:nccl`` (``nccl`` is RCCL on ROCm).
* GPU, several ranks sharing a device (functional testing on a 1-GPU box): gloo only - RCCL refuses
  duplicate devices - while the kernels still talk through hipIpc mappings.
* no GPU: gloo + the shared-memory host heap.
"""
from __future__ import annotations

import atexit
import datetime
import contextlib
import os
import threading
import random
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from .symm import SymmetricHeap


@dataclass
class DistEnv:
    rank: int
    local_rank: int
    world_size: int
    local_world_size: int
    device: torch.device
    backend: str                     # "hip" or "host"
    tp_group: Optional[dist.ProcessGroup]
    cpu_group: Optional[dist.ProcessGroup]
    heap: SymmetricHeap
    shared_gpu: bool = False         # several ranks on one device
    has_rccl: bool = False
    side_stream: Optional[object] = None   # virtual ranks: the one side stream that owns a hardware queue
    heap_mode: str = "coarse"        # memory mode of the data arena: coarse | fine | uncached | host
    selftest: Optional[object] = None      # utils.selftest.ProbeResult of the mode in use (None: not run)
    selftest_log: tuple = ()               # (mode name, description) of every mode that was probed

    @property
    def num_ranks(self) -> int:
        return self.world_size

    @property
    def safe_publish(self) -> bool:
        """True when the release-fence flavour of the GPU library is in use (``TDIST_SAFE_PUBLISH=1`` or the
        self-test's last resort): every number measured with it should say so."""
        from .. import _lib

        return self.backend == "hip" and _lib.safe_publish()


_ENV: Optional[DistEnv] = None


def init_seed(seed: int = 0) -> None:
    """Deterministic seeding (reference: python/triton_dist/utils.py:77-96)."""
    os.environ.setdefault("CUBLAS_WORKSPACE_CONFIG", ":4096:8")
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


_SCOPED = threading.local()


@contextlib.contextmanager
def use_env(env: "DistEnv"):
    """Make ``env`` what ``get_env()`` returns on this thread (a virtual cluster runs every rank's function
    inside one, so code written against the process-wide environment runs unchanged as any rank)."""
    prev = getattr(_SCOPED, "env", None)
    _SCOPED.env = env
    try:
        yield env
    finally:
        _SCOPED.env = prev


def is_initialized() -> bool:
    return getattr(_SCOPED, "env", None) is not None or _ENV is not None


def get_env() -> DistEnv:
    scoped = getattr(_SCOPED, "env", None)
    if scoped is not None:
        return scoped
    if _ENV is None:
        raise RuntimeError("call initialize_distributed() first")
    return _ENV


def initialize_distributed(seed: Optional[int] = None, heap_bytes: Optional[int] = None,
                           sig_bytes: Optional[int] = None, force_host: bool = False) -> DistEnv:
    global _ENV
    if _ENV is not None:
        return _ENV
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    world_size = int(os.environ.get("WORLD_SIZE", 1))
    local_world_size = int(os.environ.get("LOCAL_WORLD_SIZE", world_size))

    use_gpu = torch.cuda.is_available() and not force_host
    shared_gpu = False
    if use_gpu:
        ndev = torch.cuda.device_count()
        shared_gpu = local_world_size > ndev
        dev_index = local_rank % ndev
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
    else:
        device = torch.device("cpu")

    has_rccl = use_gpu and not shared_gpu
    tp_group = cpu_group = None
    if world_size > 1 or "MASTER_ADDR" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if not dist.is_initialized():
            backend = "cpu:gloo,cuda:nccl" if has_rccl else "gloo"
            kwargs = {}
            if has_rccl:
                kwargs["device_id"] = device
            dist.init_process_group(backend=backend, world_size=world_size, rank=rank,
                                    timeout=datetime.timedelta(seconds=1800), **kwargs)
        tp_group = dist.new_group(ranks=list(range(world_size)), backend="nccl" if has_rccl else "gloo")
        cpu_group = dist.new_group(ranks=list(range(world_size)), backend="gloo")
        dist.barrier(group=cpu_group)

    init_seed((seed if seed is not None else 0) + rank)
    if use_gpu:
        heap, heap_mode, probe, log = _create_checked_heap(rank, world_size, cpu_group, heap_bytes, sig_bytes)
    else:
        heap = SymmetricHeap(rank, world_size, cpu_group, "host", heap_bytes, sig_bytes)
        heap_mode, probe, log = "host", None, ()
    _ENV = DistEnv(rank, local_rank, world_size, local_world_size, device, "hip" if use_gpu else "host",
                   tp_group, cpu_group, heap, shared_gpu, has_rccl, heap_mode=heap_mode, selftest=probe,
                   selftest_log=log)
    atexit.register(finalize_distributed)
    return _ENV


def _create_checked_heap(rank: int, world_size: int, cpu_group, heap_bytes, sig_bytes):
    """Create the GPU symmetric heap and prove, before any op uses it, that payloads written by a peer are
    read correctly through this GPU's caches (utils/selftest.py).  Modes are tried in the order
    coarse-grained -> fine-grained -> uncached; the first one that passes on EVERY rank is kept.  When none does,
    the process switches to the release-fence flavour of the GPU library (``_lib.use_safe_publish``: a system-scope
    release behind every payload publish instead of relying on write-through stores) and walks the modes once
    more, so a node on which the zero-fence publish does not hold still runs - slower, and labelled
    (``DistEnv.safe_publish``).  The verdict of a probe is the same on every rank (it is all-reduced), so all ranks
    switch together.

    Environment: ``TDIST_HEAP_ALLOC_MODE=m`` pins the mode (still probed; a failure raises),
    ``TDIST_SELFTEST=0`` skips the probe, ``TDIST_SELFTEST_ROUNDS`` sets its length,
    ``TDIST_SELFTEST_FAIL_MODES=0,1`` treats the listed modes as failed with the default library (exercises the
    fall-back; ``0,1,2`` ends in the release-fence flavour) and ``TDIST_SAFE_PUBLISH=1`` starts with that flavour."""
    from .. import _lib
    from .selftest import HEAP_MODE_NAMES, coherence_probe, probe_rounds

    pinned = os.environ.get("TDIST_HEAP_ALLOC_MODE")
    modes = [int(pinned)] if pinned is not None else [0, 1, 2]
    run_probe = world_size > 1 and os.environ.get("TDIST_SELFTEST", "1") != "0"
    forced_bad = {int(x) for x in os.environ.get("TDIST_SELFTEST_FAIL_MODES", "").split(",") if x.strip()}
    log = []
    for flavour in ("default", "safe"):
        if flavour == "safe":
            if _lib.safe_publish():
                break                       # (already the flavour the first pass ran with)
            if rank == 0:
                print("[tdist] falling back to the release-fence publish (libtdist_hip_safe.so)", flush=True)
            _lib.use_safe_publish()
        tag = " +release-fence" if _lib.safe_publish() else ""
        for mode in modes:
            heap = SymmetricHeap(rank, world_size, cpu_group, "hip", heap_bytes, sig_bytes, data_mode=mode)
            if not run_probe:
                return heap, HEAP_MODE_NAMES[mode], None, ()
            res = coherence_probe(heap, cpu_group, rounds=probe_rounds())
            if mode in forced_bad and not _lib.safe_publish():
                res.ok = False
            log.append((HEAP_MODE_NAMES[mode] + tag, ("ok " if res.ok else "FAILED ") + res.describe()))
            if res.ok:
                return heap, HEAP_MODE_NAMES[mode], res, tuple(log)
            if rank == 0:
                print(f"[tdist] symmetric heap self-test FAILED in mode {HEAP_MODE_NAMES[mode]}{tag}: {res.describe()}",
                      flush=True)
            heap.close(sync_group=cpu_group)
    raise RuntimeError("symmetric heap self-test failed in every memory mode, with and without release fences: " +
                       "; ".join(f"{m}: {d}" for m, d in log))


def create_virtual_node(world_size: int, heap_bytes: int = 1 << 28, sig_bytes: int = 1 << 22,
                        backend: Optional[str] = None):
    """``world_size`` ranks hosted by THIS process (one DistEnv per virtual rank, no process
    group).  Every rank's symmetric arena is an ordinary allocation of this process, so fused
    multi-rank kernels can be exercised on one GPU by launching each rank's op on its own stream.
    Hardware queues: a rank that spins on a peer must not share a queue with it, so run with
    GPU_MAX_HW_QUEUES = min(32, 2 * world_size + 4), set before HIP initialises (at most 14 virtual
    ranks: utils/cli.py refuses more).  The variable is raised here when it is too low and torch has not touched
    the GPU yet; a process that already initialised HIP with fewer queues gets a warning - ranks that run TOGETHER
    would wait for peers queued behind them until the spin budget runs out (a probe that runs one rank of the
    cluster alone is fine)."""
    need = min(32, 2 * world_size + 4)
    have = int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)
    if backend != "host" and world_size > 1 and have < need:
        if torch.cuda.is_initialized():
            import warnings

            warnings.warn(f"{world_size} virtual ranks running together need GPU_MAX_HW_QUEUES >= {need} (it is "
                          f"{have or 'unset'}) and HIP is already initialised: export it before the first GPU call")
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = str(need)
    backend = backend or ("hip" if torch.cuda.is_available() else "host")
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "hip" else torch.device("cpu")
    heaps = SymmetricHeap.create_virtual(world_size, backend, heap_bytes, sig_bytes)
    return [DistEnv(r, r, world_size, world_size, device, backend, None, None, heaps[r], True, False)
            for r in range(world_size)]


def default_num_ctas(env: "DistEnv", num_ranks: int, requested: int = 0) -> int:
    """Workgroup budget of a persistent kernel.  0 (one workgroup per CU) is right when a rank owns
    its GPU; ranks that share one device must all stay co-resident (each spins on its peers), so
    they split the CUs between them."""
    if requested:
        return requested
    if env is not None and env.shared_gpu and env.backend == "hip" and num_ranks > 1:
        return max(8, 200 // num_ranks)
    return 0


def forbid_graph_capture(op: str, world_size: int) -> None:
    """The fused multi-rank ops number their calls with a host-side epoch (it selects the buffer
    parity and the flag value a kernel waits for).  A captured graph would replay ONE epoch forever,
    so peers' flags from the previous replay would satisfy the waits: refuse instead of racing."""
    if world_size > 1 and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise RuntimeError(f"{op}: multi-rank fused ops cannot be captured into a HIP graph (per-call epochs are "
                           "host state); run them eagerly")


def finalize_distributed() -> None:
    """Tear down heap and process groups (reference: python/triton_dist/utils.py:145-148)."""
    global _ENV
    if _ENV is None:
        return
    env, _ENV = _ENV, None
    try:
        if env.backend == "hip":
            torch.cuda.synchronize()
        if dist.is_initialized() and env.cpu_group is not None:
            dist.barrier(group=env.cpu_group)
    except Exception:
        pass
    env.heap.close()
    try:
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:
        pass
