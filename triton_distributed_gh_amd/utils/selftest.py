"""Start-up self-test of the symmetric heap: is peer-written data coherent with local caches?

Every fused op hands payloads between GPUs as "write-through stores into the consumer's arena, then an
epoch flag; the consumer polls, acquires once at system scope and reads" and reuses its buffers every
second call.  Whether the consumer's caches can still serve lines of the call before last depends on
how the arena is mapped and on the machine, so ``initialize_distributed`` measures it once with
``csrc/ops/selftest.hip`` (200 rounds, alternating parity, every word checked through plain loads and
through direct-to-LDS loads, plus fabric atomics) and moves the data arena to the next memory mode
(coarse-grained -> fine-grained -> uncached) when a single word is wrong.

The reference has no counterpart: it validates on real ranks with ``--check`` loops per op
(reference: python/triton_dist/test/amd/test_ag_gemm_intra_node.py:229-275,
test_gemm_rs_intra_node.py:229-251).
"""
from __future__ import annotations

import ctypes
import dataclasses
import os
from typing import Optional

import torch
import torch.distributed as dist

from .. import _lib

HEAP_MODE_NAMES = {0: "coarse", 1: "fine", 2: "uncached"}


@dataclasses.dataclass
class ProbeResult:
    ok: bool
    rounds: int
    plain_mismatches: int
    lds_mismatches: int
    timeouts: int
    atomic_sum: int
    atomic_expected: int
    first_bad_word: int

    def describe(self) -> str:
        return (f"rounds={self.rounds} plain_mismatches={self.plain_mismatches} lds_dma_mismatches={self.lds_mismatches} "
                f"timeouts={self.timeouts} atomics={self.atomic_sum}/{self.atomic_expected}"
                + (f" first_bad_word={self.first_bad_word}" if not self.ok else ""))


def coherence_probe(heap, cpu_group: Optional[dist.ProcessGroup], rounds: int = 200, words: int = 32768,
                    num_ctas: int = 64, stream=None) -> ProbeResult:
    """Collective over the heap's ranks (hip backend, world size >= 2).  Returns this rank's counts
    with ``ok`` already agreed across ranks (False everywhere if any rank saw a problem)."""
    assert heap.backend == "hip" and heap.world_size >= 2
    lib = _lib.hip()
    w = heap.world_size
    buf = heap.create_tensor([2 * w * words], torch.int32)
    ready = heap.create_tensor([2 * w], torch.int32, arena="sig")
    data = heap.create_tensor([2 * w], torch.int32, arena="sig")
    counter = heap.create_tensor([1], torch.int32, arena="sig")
    local = torch.zeros(8 + 16, dtype=torch.int32, device=buf.device)
    torch.cuda.synchronize()
    if cpu_group is not None:
        dist.barrier(group=cpu_group)           # every rank's buffers are zeroed before anyone writes
    cstruct = heap.ctx_struct()
    s = stream if stream is not None else torch.cuda.current_stream()
    for r in range(rounds):
        _lib.check(lib.tdist_coherence_probe(ctypes.byref(cstruct), _lib.ptr(buf), _lib.ptr(ready), _lib.ptr(data),
                                             _lib.ptr(counter), _lib.ptr(local), words, r, num_ctas,
                                             _lib.stream_handle(s)), "tdist_coherence_probe")
    torch.cuda.synchronize()
    if cpu_group is not None:
        dist.barrier(group=cpu_group)           # all peers' last atomic adds have landed
    loc = local.cpu().tolist()
    atomic_sum = int(counter.cpu()[0])
    expected = rounds * (w - 1)
    ok = loc[0] == 0 and loc[1] == 0 and loc[2] == 0 and atomic_sum == expected
    if cpu_group is not None:
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=cpu_group)
        ok = bool(flag[0])
    for t in (buf, ready, data, counter):
        heap.free_tensor(t)
    return ProbeResult(ok, rounds, loc[0], loc[1], loc[2], atomic_sum, expected, loc[4])


def probe_rounds() -> int:
    return int(os.environ.get("TDIST_SELFTEST_ROUNDS", "200"))
