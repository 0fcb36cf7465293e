"""The SHMEM API surface under the reference's names (nvshmem_* helpers of triton_dist.utils and
libshmem_device): symmetric allocation, my_pe / n_pes, putmem / getmem, putmem_signal +
signal_wait_until, int_p, broadcast, fcollect, barrier_all - on the process's own rank layout
(torchrun: one rank per GPU; plain python: world size 1, every "remote" PE is this rank).

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_nvshmem_api.py
    python triton_dist/test/nvidia/test_nvshmem_api.py

(Scenarios of reference python/triton_dist/test/nvidia/test_nvshmem_api.py; the device-side forms of
the same calls are exercised by test/amd/test_rocshmem_api.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import triton_dist.utils as U  # noqa: E402
from triton_dist.language.extra import libshmem_device as shmem  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=4096)
    args = ap.parse_args()
    U.initialize_distributed()
    rank, w, n = shmem.my_pe(), shmem.n_pes(), args.n
    sync = (lambda: torch.cuda.synchronize()) if torch.cuda.is_available() else (lambda: None)
    dev = "cuda" if torch.cuda.is_available() else "cpu"

    buf = U.nvshmem_create_tensor([w, n], torch.int32)
    sig = U.nvshmem_create_tensor([w], torch.int64)
    views = U.nvshmem_create_tensors([4], torch.float32)
    assert len(views) == w and views[rank].shape == (4,)
    assert shmem.remote_ptr(buf, rank).data_ptr() == buf.data_ptr()
    assert shmem.remote_mc_ptr(shmem.NVSHMEM_TEAM_WORLD, buf) is None          # no multicast on xGMI

    # putmem_signal to every PE's slot [rank]; wait for every PE's signal; getmem it back
    src = torch.full((n,), 100 + rank, dtype=torch.int32, device=dev)
    shmem.barrier_all()
    for pe in range(w):
        shmem.putmem_signal_block(buf[rank], src, n * 4, sig[rank:rank + 1], 1, shmem.NVSHMEM_SIGNAL_SET, pe)
    for pe in range(w):
        shmem.signal_wait_until(sig[pe:pe + 1], shmem.NVSHMEM_CMP_EQ, 1)
    sync()
    assert buf[:, 0].tolist() == [100 + s for s in range(w)]
    back = torch.zeros(n, dtype=torch.int32, device=dev)
    peer = (rank + 1) % w
    shmem.getmem_block(back, buf[peer], n * 4, peer)
    shmem.quiet()
    sync()
    assert int(back[0]) == 100 + peer

    # int_p, broadcast, fcollect
    word = U.nvshmem_create_tensor([1], torch.int32)
    shmem.barrier_all()
    shmem.int_p(word, 42 + rank, peer)
    shmem.barrier_all()
    sync()
    assert int(word[0]) == 42 + (rank - 1) % w
    bsrc = U.nvshmem_create_tensor([8], torch.float32)
    bsrc.fill_(float(rank + 1))
    bdst = torch.zeros(8, device=dev)
    gath = torch.zeros(w * 8, device=dev)
    shmem.barrier_all()
    shmem.broadcast(shmem.NVSHMEM_TEAM_WORLD, bdst, bsrc, 8, 0)
    shmem.fcollect(shmem.NVSHMEM_TEAM_WORLD, gath, bsrc, 8)
    shmem.barrier_all()
    sync()
    assert bdst.tolist() == [1.0] * 8
    assert gath.view(w, 8)[:, 0].tolist() == [float(s + 1) for s in range(w)]

    U.nvshmem_barrier_all_on_stream()
    U.nvshmem_free_tensor_sync(word)
    names = [x for x in dir(shmem) if not x.startswith("_")]
    print(f"✅ SHMEM API on {w} rank(s): put/get/signal/int_p/broadcast/fcollect/barrier ok "
          f"({len(names)} libshmem_device names)", flush=True)
    U.finalize_distributed()


if __name__ == "__main__":
    main()
