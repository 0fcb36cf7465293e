"""TCP JSON generation server on rank 0 (localhost:9999), prompt broadcast to the other ranks.

    python triton_dist/mega_triton_kernel/test/models/model_server.py --model Qwen/Qwen3-8B

(Role of reference python/triton_dist/mega_triton_kernel/test/models/model_server.py; implementation in
triton_distributed_gh_amd/mega/server.py.)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), *[".."] * 4))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = str(max(24, int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)))

from triton_distributed_gh_amd.mega import server  # noqa: E402

if __name__ == "__main__":
    sys.argv.insert(1, "serve")
    server.main()
