"""Mega-kernel attention path: qk-norm + RoPE + paged KV-cache update, split-KV attention and combine.

    python triton_dist/mega_triton_kernel/test/ops/test_page_attn.py

(Scenario of reference python/triton_dist/mega_triton_kernel/test/ops/test_page_attn.py.)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), *[".."] * 4))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = str(max(24, int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)))

from tests.test_mega import test_qk_norm_rope_kv_and_paged_attention_tasks  # noqa: E402

if __name__ == "__main__":
    for case in [(2, 8, 2, 128, 16, 300), (4, 4, 4, 64, 64, 1000), (1, 32, 8, 128, 1, 77)]:
        test_qk_norm_rope_kv_and_paged_attention_tasks(*case)
    print("✅ mega kernel qk_norm_rope_update_kvcache + paged attention match the references")
