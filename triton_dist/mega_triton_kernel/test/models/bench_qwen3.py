"""Decode-step benchmark: torch eager / torch + HIP graph / mega kernel.

    python triton_dist/mega_triton_kernel/test/models/bench_qwen3.py --model Qwen/Qwen3-8B

(Role of reference python/triton_dist/mega_triton_kernel/test/models/bench_qwen3.py; implementation in
benchmarks/bench_qwen3_mega.py.)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), *[".."] * 4))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = str(max(24, int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)))

import runpy  # noqa: E402

if __name__ == "__main__":
    runpy.run_path(os.path.join(ROOT, "benchmarks", "bench_qwen3_mega.py"), run_name="__main__")
