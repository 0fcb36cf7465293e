"""Mega-kernel allreduce and barrier tasks on virtual ranks (linear -> allreduce -> barrier per rank).

    python triton_dist/mega_triton_kernel/test/ops/test_allreduce.py [world_size]

(Scenario of reference python/triton_dist/mega_triton_kernel/test/ops/test_allreduce.py.)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), *[".."] * 4))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = str(max(24, int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)))

from tests.test_mega import test_allreduce_and_barrier_tasks_on_virtual_ranks  # noqa: E402

if __name__ == "__main__":
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    test_allreduce_and_barrier_tasks_on_virtual_ranks(w)
    print(f"✅ mega kernel allreduce + barrier tasks on {w} virtual ranks match the sum of the partials")
