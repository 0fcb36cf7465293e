"""Symmetric tensors larger than one hipIpc segment: the virtual-memory-management arena
(utils/vmm.py) shared between two processes as dmabuf file descriptors.  Scenario of the reference:
rocshmem_malloc / nvshmem_create_tensor of arbitrary size (shmem/rocshmem_bind/pyrocshmem/src/pyrocshmem.cc:101-113)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _torchrun(nproc, script, *args, port, timeout=240, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TDIST_LOG="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script), *args]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_tensor_beyond_two_gib_single_rank():
    """One symmetric tensor of 2.5 GiB (1 GiB chunks, contiguous over three of them): impossible through one
    hipIpc mapping.  In-process, so it runs on every driver."""
    import torch

    from triton_distributed_gh_amd.utils.symm import SymmetricHeap

    heap = SymmetricHeap(0, 1, None, "hip", heap_bytes=64 << 20)
    try:
        n = (2560 << 20) // 4
        t = heap.create_tensor([n], torch.int32)
        assert heap.info(t).arena == "vmm" and heap._vmm.committed == 3 << 30
        t[::4096] = 7                                   # touches every chunk, crosses both chunk borders
        edge = (1 << 30) // 4
        t[edge - 8:edge + 8] = 3
        torch.cuda.synchronize()
        assert int(t[::4096].sum()) - 7 * ((n + 4095) // 4096) in (0, -4), "strided writes"
        assert int(t[edge - 8:edge + 8].sum()) == 48
        cs = heap.ctx_struct()
        assert cs.vmm_bytes == heap._vmm.capacity and cs.vmm_base[0] == heap._vmm.bases[0]
        assert heap.translate(t.data_ptr() + 12, 0) == t.data_ptr() + 12
        small = heap.create_tensor([1024], torch.float32)          # ordinary tensors keep to the hipIpc segment
        assert heap.info(small).arena == "data"
        heap.free_tensor(t)
        t2 = heap.create_tensor([n // 2], torch.int32)              # the range is reused, nothing new is committed
        assert heap.info(t2).offset == 0 and heap._vmm.committed == 3 << 30
    finally:
        heap.close()


def test_tensor_spanning_chunks_between_processes():
    """160 MiB tensor, 64 MiB hipIpc segments, 64 MiB VMM chunks, two processes: peer writes land and the
    device-side translation of a VMM pointer works in the pull and push all-gather kernels - or, on a driver
    that cannot import another process's chunk, every rank gets the same MemoryError and the hipIpc segments
    keep serving (both outcomes are printed by the worker; a crash or a hang is the failure)."""
    res = _torchrun(2, ROOT / "tests/workers/vmm_heap_worker.py", "160", port=29581,
                    TDIST_SYMM_VMM_CHUNK_BYTES=str(64 << 20))
    assert res.returncode == 0 and ("vmm ok" in res.stdout or "vmm unsupported" in res.stdout), \
        res.stdout[-2000:] + res.stderr[-3000:]


def test_fused_op_with_every_buffer_on_the_vmm_arena():
    """TDIST_HEAP_KIND=vmm: the GEMM+RS scatter / partial buffers live on the VMM arena (or, where chunks
    cannot cross processes, the heap falls back to its hipIpc segments on every rank together)."""
    res = _torchrun(2, ROOT / "triton_dist/test/amd/test_gemm_rs_intra_node.py", "1024", "1024", "1024", "--check",
                    port=29583, TDIST_HEAP_KIND="vmm", TDIST_SYMM_VMM_CHUNK_BYTES=str(64 << 20))
    assert res.returncode == 0 and "check passed" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
