"""Device primitives one at a time on virtual ranks (reference scenarios:
test/amd/test_distributed-notify-wait.py ring queue with 16 producers + 4 consumers and many repeats,
test_wait_and_notify.py stream write / wait value, test_barrier_all.py staggered arrival,
test_rocshmem_api.py put / get / signal, test_libdevice.py scoped atomics)."""
import pytest
import torch

import triton_distributed_gh_amd.language as dl
from triton_distributed_gh_amd.ops import primitives
from triton_distributed_gh_amd.ops.common_ops import BarrierAllContext, barrier_all_on_stream, set_signal, wait_eq
from triton_distributed_gh_amd.utils.testing import VirtualCluster

pytestmark = pytest.mark.gpu


def test_scoped_atomics():
    word, out = primitives.atomics_smoke()
    torch.cuda.synchronize()
    assert out[:6].tolist() == [5, 5, 8, 11, 12, 20] and int(word) == 20


@pytest.mark.parametrize("w", [2, 4, 8])
def test_ring_queue_notify_wait(w):
    Q, elems, items, repeats = 8, 256, 96, 20
    vc = VirtualCluster(w, heap_bytes=1 << 24)
    try:
        queues = vc.collective(lambda r, env: env.heap.create_tensor([Q * elems], torch.int32))
        sigs = vc.collective(lambda r, env: env.heap.create_tensor([Q], torch.int32, arena="sig"))
        errs = [torch.zeros(4, dtype=torch.int32, device="cuda") for _ in range(w)]
        for rep in range(repeats):
            ins = [torch.randint(0, 1 << 30, (items * elems,), device="cuda", dtype=torch.int32) for _ in range(w)]
            outs = [torch.zeros(items * elems, device="cuda", dtype=torch.int32) for _ in range(w)]
            for s in sigs:
                s.zero_()
            torch.cuda.synchronize()
            vc.run(lambda r, env: primitives.ring_queue(queues[r], sigs[r], ins[r], outs[r], (r + 1) % w, Q, elems, items,
                                                        num_producers=8, num_consumers=4, err=errs[r], env=env))
            for r in range(w):
                assert int(errs[r][0]) == 0
                assert torch.equal(outs[(r + 1) % w], ins[r]), f"repeat {rep}: rank {(r + 1) % w} got wrong data"
    finally:
        vc.close()


def test_wait_kernel_and_notify_across_virtual_ranks():
    vc = VirtualCluster(2, heap_bytes=1 << 22)
    try:
        flags = vc.collective(lambda r, env: env.heap.create_tensor([4], torch.int32, arena="sig"))
        status = {}

        def fn(r, env):
            if r == 1:
                status[1] = primitives.wait_flag(flags[1], 7)            # blocks stream 1 until rank 0 notifies
            else:
                torch.cuda._sleep(5_000_000)
                dl.notify(flags[0], 1, signal=7, sig_op="set", comm_scope="intra_node", env=env)

        vc.run(fn)
        assert int(status[1][0]) == 1 and int(flags[1][0]) == 7
        # "add" semantics and the stream-ordered wait of the host API
        vc.run(lambda r, env: dl.notify(flags[r], 1 - r, signal=3, sig_op="add", env=env))
        assert int(flags[0][0]) == 3 and int(flags[1][0]) == 10
        vc.run(lambda r, env: dl.wait(flags[r], 1, "sys", "acquire", waitValue=int(flags[r][0]), env=env))
    finally:
        vc.close()


def test_stream_write_and_wait_value():
    vc = VirtualCluster(2, heap_bytes=1 << 20)       # for its two streams on separate hardware queues
    s1, s2 = vc.streams
    sig = torch.zeros(2, dtype=torch.int64, device="cuda")
    out = torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(s2):
        wait_eq(sig[0:1], 5, s2)
        out.fill_(1.0)                       # ordered after the wait
    assert float(out) == 0.0
    with torch.cuda.stream(s1):
        set_signal(sig[0:1], 5, s1)
    torch.cuda.synchronize()
    assert float(out) == 1.0 and int(sig[0]) == 5
    vc.close()


@pytest.mark.parametrize("w", [2, 4, 8])
def test_barrier_all_with_staggered_arrival(w):
    vc = VirtualCluster(w, heap_bytes=1 << 22)
    try:
        ctxs = vc.collective(lambda r, env: BarrierAllContext(env=env))
        marks = [torch.zeros(w, device="cuda") for _ in range(w)]
        for it in range(5):
            def fn(r, env):
                torch.cuda._sleep(2_000_000 * ((r + it) % w))     # every rank arrives at a different time
                marks[r][r] = it + 1
                barrier_all_on_stream(ctxs[r])
                return torch.stack([m[i] for i, m in enumerate(marks)]).clone()   # everyone's mark, read after the barrier

            outs = vc.run(fn)
            for r in range(w):
                assert outs[r].tolist() == [it + 1] * w, f"rank {r} passed the barrier early"
    finally:
        vc.close()


@pytest.mark.parametrize("w", [2, 4])
def test_shmem_put_get_signal(w):
    n = 1024
    vc = VirtualCluster(w, heap_bytes=1 << 22)
    try:
        bufs = vc.collective(lambda r, env: env.heap.create_tensor([w * n], torch.int32))
        sigs = vc.collective(lambda r, env: env.heap.create_tensor([w], torch.int64, arena="sig"))
        srcs = [torch.full((n,), 100 + r, dtype=torch.int32, device="cuda") for r in range(w)]
        backs = [torch.zeros(w * n, dtype=torch.int32, device="cuda") for _ in range(w)]
        torch.cuda.synchronize()
        vc.run(lambda r, env: primitives.shmem_api_smoke(bufs[r], sigs[r], srcs[r], backs[r], n, 77, env=env))
        for r in range(w):
            assert bufs[r].view(w, n)[:, 0].tolist() == [100 + s for s in range(w)]     # slot s written by rank s
            assert sigs[r].tolist() == [77] * w
            assert backs[r].view(w, n)[:, 0].tolist() == [100 + r] * w                  # our slot read back from every peer
    finally:
        vc.close()


@pytest.mark.parametrize("blocks", [8, 256, 1024])
def test_grid_barrier(blocks):
    from triton_distributed_gh_amd.ops.common_ops import barrier_on_this_grid_smoke
    stale, timeouts = barrier_on_this_grid_smoke(blocks, rounds=16)
    assert (stale, timeouts) == (0, 0)


@pytest.mark.parametrize("w", [1, 4])
def test_language_extra_helpers(w):
    """tdist/language_extra.hpp (the gfx950 forms of the reference's language_extra.py:109-929): packed moves,
    vector ld / st, 64-lane shuffles and ballot, one-atomic-per-wave, multimem stores to every rank, arrive_inc /
    wait_eq across ranks, scoped ld / st / CAS / red - one kernel per rank checks itself; then the load-reduce
    over the ranks against a PyTorch sum."""
    vc = VirtualCluster(w, heap_bytes=1 << 24)
    try:
        bufs = vc.collective(lambda r, env: env.heap.create_tensor([(2 * w + 2) * 512], torch.int32))
        syncs = vc.collective(lambda r, env: env.heap.create_tensor([4], torch.int32, arena="sig"))
        outs = [torch.zeros(4, dtype=torch.int32, device="cuda") for _ in range(w)]
        torch.cuda.synchronize()
        vc.run(lambda r, env: primitives.language_extra_smoke(bufs[r], syncs[r], outs[r], env=env))
        for r in range(w):
            assert outs[r].tolist() == [1, 1, 1, 0], f"rank {r}: wave 0 / wave 1 self-checks, arrival, failed-check bits = {outs[r].tolist()}"
            assert int(syncs[r][0]) == w
        srcs = vc.collective(lambda r, env: env.heap.create_tensor([128 * 8], torch.bfloat16))
        vals = [torch.randn(128 * 8, device="cuda").bfloat16() for _ in range(w)]
        for r in range(w):
            srcs[r].copy_(vals[r])
        dsts = [torch.empty(128 * 8, device="cuda", dtype=torch.bfloat16) for _ in range(w)]
        torch.cuda.synchronize()
        vc.run(lambda r, env: primitives.ld_reduce_smoke(srcs[r], dsts[r], env=env))
        ref = sum(v.float() for v in vals).bfloat16()
        for r in range(w):
            assert torch.equal(dsts[r], ref)
    finally:
        vc.close()
