"""Runtime utilities: perf_func, dist_print, assert_allclose, generate_data, sleep_async, topology
and hardware probes, perf models, group_profile.

    python triton_dist/test/nvidia/test_utils.py

(Scenario of reference python/triton_dist/test/nvidia/test_utils.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import triton_dist.utils as U  # noqa: E402


def main():
    argparse.ArgumentParser().parse_args()
    U.initialize_distributed()
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    a = torch.rand(256, 256, device=dev)
    out, ms = U.perf_func(lambda: a @ a, iters=5, warmup_iters=2)
    assert out.shape == (256, 256) and ms > 0
    U.assert_allclose(a, a.clone(), rtol=0, atol=0)
    try:
        U.assert_allclose(a, a + 1, rtol=1e-3, atol=1e-3, verbose=False)
        raise SystemExit("assert_allclose accepted different tensors")
    except (AssertionError, RuntimeError):
        pass
    U.dist_print("dist_print from rank 0", allowed_ranks=[0])
    assert U.is_hip() == torch.cuda.is_available() and not U.is_cuda()
    assert isinstance(U.has_fullmesh_nvlink(), bool) and not U.is_nvshmem_multimem_supported() and not U.has_tma()
    assert U.get_numa_world_size() >= 1
    if torch.cuda.is_available():
        U.sleep_async(1.0)
        torch.cuda.synchronize()
        assert U.get_device_max_shared_memory_size() >= 64 * 1024
    from triton_dist.kernels.nvidia.comm_perf_model import estimate_all_gather_time_ms, estimate_reduce_scatter_time_ms
    from triton_dist.kernels.nvidia.gemm_perf_model import estimate_gemm_sol_time_ms

    assert estimate_gemm_sol_time_ms(4096, 8192, 8192) > 0
    assert estimate_all_gather_time_ms(8 << 20, 8) > 0 and estimate_reduce_scatter_time_ms(64 << 20, 8) > 0
    with U.group_profile("utils_smoke", do_prof=False):
        (a @ a).sum().item()
    print("✅ utils: perf_func, assert_allclose, dist_print, probes, perf models, group_profile ok", flush=True)
    U.finalize_distributed()


if __name__ == "__main__":
    main()
