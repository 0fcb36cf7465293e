"""Mega-kernel element-wise tasks (add, silu*up) next to rms_norm, one persistent launch, vs PyTorch.

    python triton_dist/mega_triton_kernel/test/ops/test_add.py

(Scenario of reference python/triton_dist/mega_triton_kernel/test/ops/test_add.py.)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), *[".."] * 4))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = str(max(24, int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)))

from tests.test_mega import test_norm_add_silu_tasks  # noqa: E402

if __name__ == "__main__":
    test_norm_add_silu_tasks()
    print("✅ mega kernel add / silu_mul_up / rms_norm tasks match PyTorch")
