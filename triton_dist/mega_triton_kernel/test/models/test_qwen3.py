"""Whole-model check: one decode step of the mega kernel vs the layer-by-layer torch forward.

    python triton_dist/mega_triton_kernel/test/models/test_qwen3.py [world_size] [interpreter|graph|auto]

(Scenario of reference python/triton_dist/mega_triton_kernel/test/models/test_qwen3.py.)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), *[".."] * 4))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = str(max(24, int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)))

from tests.test_mega import test_qwen3_mega_forward_matches_torch_decode  # noqa: E402

if __name__ == "__main__":
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    backend = sys.argv[2] if len(sys.argv) > 2 else "interpreter"
    # (the graph back-end's attention lowering wants head size 128)
    name = "tdist/tiny-dense-d128" if backend == "graph" else "tdist/tiny-dense"
    test_qwen3_mega_forward_matches_torch_decode(w, name, backend)
    print(f"✅ Qwen3 mega forward ({backend} back-end) matches the torch decode step (world size {w})")
