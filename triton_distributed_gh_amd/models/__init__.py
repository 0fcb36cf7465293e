"""Model zoo entry points (reference: python/triton_dist/models/__init__.py:25-64)."""
from .config import KNOWN_ARCHS, ArchConfig, ModelConfig  # noqa: F401
from .dense import DenseLLM
from .kv_cache import KV_Cache, PagedKVCache  # noqa: F401
from .qwen_moe import Qwen3MoE


class AutoLLM:
    model_mapping = {
        "Qwen/Qwen3-0.6B": DenseLLM,
        "Qwen/Qwen3-1.7B": DenseLLM,
        "Qwen/Qwen3-4B": DenseLLM,
        "Qwen/Qwen3-8B": DenseLLM,
        "Qwen/Qwen3-14B": DenseLLM,
        "Qwen/Qwen3-32B": DenseLLM,
        "Qwen/Qwen3-30B-A3B": Qwen3MoE,
        "Qwen/Qwen3-235B-A22B": Qwen3MoE,
        "meta-llama/Meta-Llama-3-70B": DenseLLM,
        "ByteDance-Seed/Seed-OSS-36B-Instruct": DenseLLM,
        "tdist/tiny-dense": DenseLLM,
        "tdist/tiny-dense-d128": DenseLLM,
        "tdist/tiny-moe": Qwen3MoE,
    }

    @staticmethod
    def from_pretrained(config: ModelConfig, group=None, env=None):
        cls = AutoLLM.model_mapping.get(config.model_name)
        if cls is None:
            print(f"Model {config.model_name} not found in model mapping, available models: "
                  f"{list(AutoLLM.model_mapping)}; falling back to DenseLLM.")
            cls = DenseLLM
        return cls(config, group, env=env)


class _ByteTokenizer:
    """Offline stand-in used when no HuggingFace tokenizer is cached locally: UTF-8 bytes as ids."""

    eos_token_id = 0

    def __call__(self, text, return_tensors=None, **_):
        import torch

        texts = [text] if isinstance(text, str) else list(text)
        rows = [list(t.encode("utf-8")) or [0] for t in texts]
        n = max(len(r) for r in rows)
        ids = [[0] * (n - len(r)) + r for r in rows]
        return {"input_ids": torch.tensor(ids, dtype=torch.long)} if return_tensors == "pt" else {"input_ids": ids}

    def encode(self, text, **_):
        return list(text.encode("utf-8"))

    def decode(self, ids, skip_special_tokens=True):
        return bytes(int(i) % 256 for i in ids).decode("utf-8", errors="replace")

    def batch_decode(self, batch, skip_special_tokens=True):
        return [self.decode(row.tolist() if hasattr(row, "tolist") else row) for row in batch]


class AutoTokenizer:

    @staticmethod
    def from_pretrained(model_config):
        try:
            from transformers import AutoTokenizer as HFTokenizer

            return HFTokenizer.from_pretrained(model_config.model_name, use_fast=True, legacy=False,
                                               local_files_only=True)
        except Exception:
            return _ByteTokenizer()


from .engine import Engine  # noqa: E402,F401
