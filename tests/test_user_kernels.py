"""The user-kernel path: a stand-alone .hip file that includes only tdist/primitives.hpp is compiled with
hipcc, loaded as a code object and launched from Python with a SymmCtx argument.

Scenario of the reference's primitive test and tutorial (reference:
python/triton_dist/test/amd/test_distributed-notify-wait.py:39-212 - ring queue, 16 producers + 4
consumers, 2000 repeats; tutorials/01-distributed-notify-wait.py:62-145) and of its AOT test
(python/triton_dist/test/nvidia/test_compile_aot.py:162-236: build, load, call, compare)."""
import ctypes
import json
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from triton_distributed_gh_amd.tools import runtime as rt

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tutorials" / "kernels" / "ring_queue_user.hip"


def test_marshal_args_builds_the_launch_array():
    """hipModuleLaunchKernel convention: element i of the array points at the value of parameter i."""
    t = torch.zeros(4)
    s = ctypes.c_longlong(1 << 40)
    arr, keep = rt.marshal_args([t, 7, True, 2.5, None, s, rt.f64(1.25)])
    at = lambda i, ct: ctypes.cast(arr[i], ctypes.POINTER(ct)).contents.value   # noqa: E731
    assert at(0, ctypes.c_void_p) == t.data_ptr()
    assert at(1, ctypes.c_int) == 7 and at(2, ctypes.c_int) == 1
    assert at(3, ctypes.c_float) == 2.5
    assert at(4, ctypes.c_void_p) in (None, 0)
    assert at(5, ctypes.c_longlong) == 1 << 40 and at(6, ctypes.c_double) == 1.25
    with pytest.raises(TypeError):
        rt.marshal_args([1 << 40])          # does not fit an int: must be wrapped explicitly
    with pytest.raises(TypeError):
        rt.marshal_args(["text"])


def test_user_kernel_compiles_against_the_public_header(tmp_path):
    """hipcc cross-compiles the tutorial kernel for gfx950 without a GPU; an unchanged source is not rebuilt."""
    if rt.hipcc_path() is None:
        pytest.skip("hipcc not installed")
    out = rt.compile_hip(SRC, tmp_path / "rq.hsaco")
    assert out.exists() and out.stat().st_size > 1000
    stamp = out.stat().st_mtime_ns
    assert rt.compile_hip(SRC, tmp_path / "rq.hsaco").stat().st_mtime_ns == stamp


def test_compile_aot_cli_builds_a_user_code_object(tmp_path):
    if rt.hipcc_path() is None:
        pytest.skip("hipcc not installed")
    ws = tmp_path / "ws"
    res = subprocess.run([sys.executable, "-m", "triton_distributed_gh_amd.tools.compile_aot", "--workspace", str(ws),
                          "--kernels", f"{SRC}:ring_queue_user", "--library", "user_kernels", "--build"],
                         cwd=ROOT, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    man = json.loads((ws / "manifest.json").read_text())
    assert man["code_objects"][0]["kernel"] == "ring_queue_user" and Path(man["code_objects"][0]["hsaco"]).exists()


@pytest.mark.gpu
def test_user_ring_queue_on_four_virtual_ranks():
    """2000 queue rounds (4 launches x 500 rounds over the same signals) between neighbours on 4 virtual
    ranks, 16 producer + 4 consumer workgroups per rank, with the kernel loaded from the code object."""
    from triton_distributed_gh_amd.utils.testing import VirtualCluster

    k = rt.load_module(rt.compile_hip(SRC)).get("ring_queue_user")
    w, Q, E, rounds, launches = 4, 8, 64, 500, 4
    N = Q * rounds
    vc = VirtualCluster(w, heap_bytes=1 << 24)
    try:
        queues = vc.collective(lambda r, env: env.heap.create_tensor([Q * E], torch.int32))
        sigs = vc.collective(lambda r, env: env.heap.create_tensor([Q], torch.int32, arena="sig"))
        errs = [torch.zeros(4, dtype=torch.int32, device="cuda") for _ in range(w)]
        for it in range(launches):
            ins = [torch.randint(-2 ** 31, 2 ** 31 - 1, (N * E,), dtype=torch.int32, device="cuda") for _ in range(w)]
            outs = [torch.zeros(N * E, dtype=torch.int32, device="cuda") for _ in range(w)]
            torch.cuda.synchronize()

            def launch(r, env):
                k(20, 256, torch.cuda.current_stream(), env.heap, queues[r], sigs[r], ins[r], outs[r], (r + 1) % w,
                  Q, E, N, it * rounds, 16, 4, errs[r])

            vc.run(launch)
            for r in range(w):
                assert int(errs[r][0]) == 0, f"rank {r}: a wait timed out"
                assert torch.equal(outs[r], ins[(r - 1) % w]), f"launch {it}, rank {r}: wrong items"
    finally:
        vc.close()


def test_extern_device_library_links_into_the_user_kernel(tmp_path):
    """The device-side form of extern_call: a kernel file that only DECLARES two device functions, built
    together with the library that defines them (cross-compiles without a GPU)."""
    if rt.hipcc_path() is None:
        pytest.skip("no hipcc")
    k = ROOT / "tutorials/kernels/extern_call_user.hip"
    lib = ROOT / "tutorials/kernels/extern_lib_user.hip"
    with pytest.raises(RuntimeError):                       # without the library the symbols stay undefined
        rt.compile_hip(k, tmp_path / "alone.hsaco")
    co = rt.compile_hip(k, tmp_path / "with_lib.hsaco", extern_libs=[lib])
    assert co.exists() and co.stat().st_size > 1000 and co.read_bytes()[:4] == b"\x7fELF"
    assert rt.compile_hip(k, tmp_path / "with_lib.hsaco", extern_libs=[lib]) == co      # digest hit, no rebuild


@pytest.mark.gpu
def test_extern_device_library_kernel_runs(tmp_path):
    from triton_distributed_gh_amd.utils.symm import SymmetricHeap

    co = rt.compile_hip(ROOT / "tutorials/kernels/extern_call_user.hip", tmp_path / "ext.hsaco",
                        extern_libs=[ROOT / "tutorials/kernels/extern_lib_user.hip"])
    kern = rt.load_module(co).get("silu_and_tag")
    heap = SymmetricHeap(0, 1, None, "hip", heap_bytes=1 << 22)
    try:
        n = 1000
        x = torch.randn(n, device="cuda")
        out = torch.empty(n, device="cuda")
        tags = torch.empty(n, dtype=torch.int32, device="cuda")
        kern((n + 255) // 256, 256, None, heap, x, out, tags, n)
        torch.cuda.synchronize()
        # resident workgroups per compute unit: what a kernel whose workgroups wait for each other sizes its grid by
        from triton_dist.utils import cuda_occupancy_max_activate_blocks_per_multiprocessor as occ

        assert 1 <= occ(kern, 4) <= 8 and occ(kern, 4, 64 << 10) <= 2
        torch.testing.assert_close(out, torch.nn.functional.silu(x), atol=1e-5, rtol=1e-5)
        assert tags.tolist() == [i * 31 for i in range(n)]
    finally:
        heap.close()
