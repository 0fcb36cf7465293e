"""CPU-only plumbing: symmetric heap, notify / wait, barrier and the fused-op protocols on the host
backend (the "world_size=1 notify/wait + reference GEMM correctness on CPU" configuration, plus
world_size 2..4 through threads and through real gloo processes)."""
import ctypes
import os
import socket
import subprocess
import sys
import threading
import time
from pathlib import Path

import pytest
import torch

from triton_distributed_gh_amd import _lib
from triton_distributed_gh_amd.ops.ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context
from triton_distributed_gh_amd.ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node
from triton_distributed_gh_amd.utils.symm import SymmetricHeap, _FreeList
from triton_distributed_gh_amd.utils.testing import VirtualCluster

ROOT = Path(__file__).resolve().parent.parent


def test_freelist_alloc_release_coalesce():
    fl = _FreeList(1024)
    a = fl.alloc(100, 64)
    b = fl.alloc(100, 64)
    c = fl.alloc(100, 64)
    assert (a, b, c) == (0, 128, 256)
    fl.release(b)
    assert fl.alloc(64, 64) == 128            # hole reused first-fit
    fl.release(a)
    fl.release(c)
    with pytest.raises(MemoryError):
        fl.alloc(4096, 64)


def test_world1_heap_and_signals():
    heap = SymmetricHeap(0, 1, None, "host", heap_bytes=1 << 20, sig_bytes=1 << 16)
    t = heap.create_tensor([4, 8], torch.float32)
    assert t.shape == (4, 8) and float(t.abs().sum()) == 0
    views = heap.peer_views(t)
    assert len(views) == 1 and views[0].data_ptr() == t.data_ptr()
    sig = heap.create_tensor([8], torch.int32, arena="sig")
    lib = _lib.host()
    lib.tdh_notify32(ctypes.c_void_p(sig.data_ptr()), 5, 1)
    lib.tdh_notify32(ctypes.c_void_p(sig.data_ptr()), 2, 2)
    assert int(sig[0]) == 7
    assert lib.tdh_wait32(ctypes.c_void_p(sig.data_ptr()), 1, 0, 7, ctypes.c_int64(100)) == 0
    assert lib.tdh_wait32(ctypes.c_void_p(sig.data_ptr()), 1, 0, 8, ctypes.c_int64(50)) == 1   # timeout
    c = heap.ctx_struct()
    assert c.rank == 0 and c.num_ranks == 1 and c.heap_base[0] == heap.base("data", 0)
    heap.free_tensor(t)
    t2 = heap.create_tensor([4, 8], torch.float32)
    assert t2.data_ptr() == t.data_ptr()
    heap.close()


def test_notify_wait_across_threads():
    vc = VirtualCluster(2, heap_bytes=1 << 20, backend="host")
    lib = _lib.host()
    sig = vc.collective(lambda r, env: env.heap.create_tensor_list([4], torch.int32, arena="sig"))
    data = vc.collective(lambda r, env: env.heap.create_tensor_list([16], torch.float32))

    def fn(r, env):
        if r == 0:
            time.sleep(0.05)
            data[0][1].fill_(3.5)                                        # store into rank 1's copy
            lib.tdh_notify32(ctypes.c_void_p(sig[0][1].data_ptr()), 1, 1)   # then signal rank 1
            return None
        rc = lib.tdh_wait32(ctypes.c_void_p(sig[1][1].data_ptr()), 1, 0, 1, ctypes.c_int64(5000))
        assert rc == 0
        return float(data[1][1].sum())

    outs = vc.run(fn)
    assert outs[1] == 3.5 * 16
    vc.close()


@pytest.mark.parametrize("w", [1, 2, 4])
def test_ag_gemm_host_protocol(w):
    M, N, K = 64 * w, 32 * w, 48
    vc = VirtualCluster(w, heap_bytes=1 << 22, backend="host")
    ctxs = vc.collective(lambda r, env: create_ag_gemm_intra_node_context(
        M, N // w, K, torch.float32, torch.float32, r, w, M_PER_CHUNK=16, env=env))
    for it in range(3):
        a = [torch.randn(M // w, K) for _ in range(w)]
        b = [torch.randn(N // w, K) for _ in range(w)]
        outs = vc.run(lambda r, env: ag_gemm_intra_node(a[r], b[r], False, ctxs[r]))
        full = torch.cat(a)
        for r in range(w):
            assert torch.allclose(outs[r], full @ b[r].t(), atol=1e-4)
    vc.close()


@pytest.mark.parametrize("w", [1, 2, 4])
def test_gemm_rs_host_protocol(w):
    M, N, K = 32 * w, 40, 24 * w
    vc = VirtualCluster(w, heap_bytes=1 << 22, backend="host")
    ctxs = vc.collective(lambda r, env: create_gemm_rs_intra_node_context(M, N, torch.float32, r, w, env=env))
    for it in range(3):
        a = [torch.randn(M, K // w) for _ in range(w)]
        b = [torch.randn(N, K // w) for _ in range(w)]
        outs = vc.run(lambda r, env: gemm_rs_intra_node(a[r], b[r], ctxs[r]))
        full = sum(a[r] @ b[r].t() for r in range(w))
        Mr = M // w
        for r in range(w):
            assert torch.allclose(outs[r], full[r * Mr:(r + 1) * Mr], atol=1e-4)
    vc.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = r"""
import sys, torch
sys.path.insert(0, %r)
from triton_distributed_gh_amd.utils import initialize_distributed, finalize_distributed
from triton_distributed_gh_amd.ops.ag_gemm import create_ag_gemm_intra_node_context, ag_gemm_intra_node
from triton_distributed_gh_amd.ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node
import torch.distributed as dist
env = initialize_distributed(seed=1, heap_bytes=1 << 22, force_host=True)
r, w = env.rank, env.world_size
M, N, K = 32 * w, 16 * w, 24
ctx = create_ag_gemm_intra_node_context(M, N // w, K, torch.float32, torch.float32, r, w, M_PER_CHUNK=8)
a, b = torch.randn(M // w, K), torch.randn(N // w, K)
c = ag_gemm_intra_node(a, b, False, ctx)
parts = [torch.empty_like(a) for _ in range(w)]
dist.all_gather(parts, a, group=env.cpu_group)
assert torch.allclose(c, torch.cat(parts) @ b.t(), atol=1e-4)
ctx2 = create_gemm_rs_intra_node_context(M, N, torch.float32, r, w)
a2, b2 = torch.randn(M, K), torch.randn(N, K)
out = gemm_rs_intra_node(a2, b2, ctx2)
full = a2 @ b2.t()
dist.all_reduce(full, group=env.cpu_group)
assert torch.allclose(out, full[r * (M // w):(r + 1) * (M // w)], atol=1e-4)
# heap growth across processes: 4 MiB segments, three 3 MiB tensors -> two extra segments
big = [env.heap.create_tensor([3 << 20], torch.uint8) for _ in range(3)]
assert [env.heap.info(t).arena for t in big][1:] == ["data1", "data2"] or env.heap.info(big[0]).arena != "data"
big[2].fill_(10 + r)
dist.barrier(group=env.cpu_group)
assert int(env.heap.peer_view(big[2], (r + 1) %% w)[123]) == 10 + (r + 1) %% w
dist.barrier(group=env.cpu_group)
finalize_distributed()
print("WORKER_OK", r)
"""


def test_two_process_gloo_shared_memory_heap(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % str(ROOT))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"WORKER_OK {r}" in o, o[-2000:]


@pytest.mark.parametrize("ntokens,E,topk,w,rank", [(64, 8, 2, 1, 0), (512, 8, 2, 4, 1), (2400, 32, 4, 8, 5)])
def test_cpp_moe_align_oracle_matches_python_reference(ntokens, E, topk, w, rank):
    from triton_distributed_gh_amd.ops.moe import _align_cpu, moe_align_reference

    torch.manual_seed(ntokens)
    ids = torch.stack([torch.randperm(E)[:topk] for _ in range(ntokens)]).int()
    tpr = ntokens // w
    got = moe_align_reference(ids, E, w, rank, tpr)
    ref = _align_cpu(ids, E, w, rank, tpr, got.sorted_ids.numel())
    nb = int(ref.meta[0])
    assert torch.equal(got.meta, ref.meta)
    assert torch.equal(got.sorted_ids, ref.sorted_ids)
    assert torch.equal(got.expert_of_block[:nb], ref.expert_of_block[:nb])
    assert torch.equal(got.seg_range[:nb], ref.seg_range[:nb])
    assert torch.equal(got.order[:nb], ref.order[:nb])


def test_symmetric_heap_grows_by_segments():
    """A heap of 1 MiB segments: the third 600 KiB tensor does not fit segments 0 / 1 and lands in a
    new one; peer views, pointer translation and the device context follow it."""
    vc = VirtualCluster(2, heap_bytes=1 << 20, backend="host")
    try:
        ts = [vc.collective(lambda r, env: env.heap.create_tensor([600 * 1024], torch.uint8)) for _ in range(3)]
        heaps = [e.heap for e in vc.envs]
        assert [heaps[0].info(t[0]).arena for t in ts] == ["data", "data1", "data2"]
        for r in range(2):
            ts[2][r].fill_(r + 7)
        assert int(heaps[0].peer_view(ts[2][0], 1)[5]) == 8 and int(heaps[1].peer_view(ts[2][1], 0)[5]) == 7
        c = heaps[0].ctx_struct()
        assert c.num_extra == 2 and c.extra_base[1][1] == heaps[0].peer_ptr(ts[2][0], 1)
        assert heaps[0].translate(ts[1][0].data_ptr() + 16, 1) == heaps[0].peer_ptr(ts[1][0], 1) + 16
        with pytest.raises(MemoryError):
            vc.collective(lambda r, env: env.heap.create_tensor([2 << 20], torch.uint8))     # larger than a segment
    finally:
        vc.close()


def test_perf_model_tables_and_estimates():
    """MI355X tables (dense MFMA peaks, HBM3E, xGMI links) and what the estimates built on them must obey
    (reference: kernels/nvidia/gemm_perf_model.py:232-236, comm_perf_model.py:94-116)."""
    import torch

    from triton_distributed_gh_amd.utils import perf_model as pm

    assert pm.get_max_tensorcore_tflops(torch.bfloat16) == 2500.0 and pm.get_max_tensorcore_tflops(torch.float8_e4m3fn) == 5000.0
    assert pm.get_dram_gbps() == 8000.0 and pm.XGMI_LINKS_PER_GPU == 7
    # headline shape: compute-bound, and the expected time of the tile engine sits above the speed of light
    sol = pm.estimate_gemm_sol_time_ms(4096, 8192, 8192)
    exp = pm.estimate_gemm_time_ms(4096, 8192, 8192)
    assert abs(sol - 2 * 4096 * 8192 * 8192 / 2.5e15 * 1e3) < 1e-6 and sol < exp < 2 * sol
    # decode shape: memory-bound
    assert pm.estimate_gemm_sol_time_ms(1, 4096, 12288) == pytest.approx(4096 * 12288 * 2 / 8e12 * 1e3, rel=1e-2)
    # a full-mesh all-gather is bound by ONE link per shard: twice the ranks, half the shard, half the time
    ag = [pm.estimate_all_gather_time_ms(4096 // w * 8192 * 2, w) for w in (2, 4, 8)]
    assert ag[0] == pytest.approx(2 * ag[1]) and ag[1] == pytest.approx(2 * ag[2])
    for w in (2, 4, 8):
        est = pm.estimate_ag_gemm_time_ms(4096, 8192 // w, 8192, w, push_wgs=pm.suggest_push_workgroups(4096, 8192 // w, 8192, w))
        assert est["total_ms"] >= max(est["gemm_ms"], est["all_gather_ms"])
        rs = pm.estimate_gemm_rs_time_ms(4096, 8192, 8192 // w, w)
        assert rs["total_ms"] >= max(rs["gemm_ms"], rs["reduce_scatter_ms"])
        n = pm.suggest_push_workgroups(4096, 8192 // w, 8192, w)
        assert n % (w - 1) == 0 and w - 1 <= n <= 64
    assert pm.suggest_push_workgroups(4096, 8192, 8192, 1) == 0
    # a tiny GEMM next to a large shard cannot be fed by a quarter of the CUs: the copy engines take over
    assert pm.prefer_copy_engine_producer(8192, 256, 8192, 8) and not pm.prefer_copy_engine_producer(4096, 8192, 8192, 2)
    # Qwen3-8B decode step: 15.1 GB of weights, ~290 launches -> between the streaming bound and twice it
    step = pm.estimate_decode_step_ms(int(15.1e9), 36 * 2 * 512 * 8 * 128 * 2, 290)
    assert 15.1e9 / 8e12 * 1e3 < step < 2 * 15.1e9 / 5.5e12 * 1e3


def test_decode_split_count_rules():
    """Split count of the decode attention grid (ops/flash_decode.py: auto_kv_splits): never more workgroups than two
    rounds of the chip (the latency form of the kernel keeps two per CU) unless a single split already exceeds it,
    at least 128 tokens (two launches) / 64 tokens (one launch) per split, more splits for longer caches, and the
    measured optima at the shapes the rule was fitted on."""
    from triton_distributed_gh_amd.ops.flash_decode import auto_kv_splits

    for fused in (False, True):
        for batch in (1, 2, 4, 8, 16, 32, 64, 200):
            prev = 0
            for L in (64, 128, 512, 576, 1000, 2048, 4096, 8192, 32768, 131072):
                s = auto_kv_splits(batch, 32, 8, L, fused=fused)
                assert 1 <= s <= 32
                assert s == 1 or batch * 8 * s <= 512, (fused, batch, L, s)
                assert s == 1 or L // s >= (64 if fused else 128), (fused, batch, L, s)
                assert s >= prev, "longer caches never get fewer splits"
                prev = s
    # group of 32 query heads per KV head: two head blocks per KV head count twice
    assert auto_kv_splits(1, 64, 2, 8192) * 1 * 2 * 2 <= 512
    # fitted shapes (two launches / one launch)
    assert auto_kv_splits(1, 32, 8, 4096) == 16 and auto_kv_splits(1, 32, 8, 512) == 4 and auto_kv_splits(1, 32, 8, 32768) == 32
    assert auto_kv_splits(4, 32, 8, 8192) == 16 and auto_kv_splits(16, 32, 8, 4096) == 4
    assert auto_kv_splits(64, 32, 8, 4096) == 1
    assert auto_kv_splits(1, 32, 8, 4096, fused=True) == 16 and auto_kv_splits(1, 32, 8, 32768, fused=True) == 32
    assert auto_kv_splits(8, 32, 8, 2048, fused=True) == 8
