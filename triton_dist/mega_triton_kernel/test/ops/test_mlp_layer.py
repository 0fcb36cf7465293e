"""Mega-kernel linear tasks: plain linear, linear + residual, fc1 (gate/up GEMM with fused SiLU*up), for
several (tokens, N, K, queue count) combinations.

    python triton_dist/mega_triton_kernel/test/ops/test_mlp_layer.py

(Scenario of reference python/triton_dist/mega_triton_kernel/test/ops/test_mlp_layer.py.)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), *[".."] * 4))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = str(max(24, int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)))

from tests.test_mega import test_linear_task_modes  # noqa: E402

if __name__ == "__main__":
    for case in [(1, 512, 256, 64), (5, 1536, 1024, 256), (16, 4096, 512, 32), (3, 200, 96, 16)]:
        test_linear_task_modes(*case)
    print("✅ mega kernel linear / fc1 / residual tasks match PyTorch")
