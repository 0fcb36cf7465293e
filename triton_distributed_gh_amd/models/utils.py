"""Logging, seeding, sampling and weight loading helpers for the model layer
(reference: python/triton_dist/models/utils.py:45-128)."""
from __future__ import annotations

import logging
import os
import random
from datetime import datetime

import numpy as np
import torch

from .weights import load_weight_source

PLATFORM = "amd"
_COLORS = {"info": "\033[36m", "warning": "\033[33m", "error": "\033[31m", "success": "\033[32m", "debug": "\033[35m"}


class MyLogger:
    """Timestamped, colourised logger; ``TRITON_DIST_DEBUG=1`` enables the debug level."""

    def __init__(self):
        self.logger = logging.getLogger("triton_dist.models")
        self.logger.setLevel(logging.INFO)
        if not self.logger.handlers:
            self.logger.addHandler(logging.StreamHandler())
        self.logger.propagate = False
        if os.getenv("TRITON_DIST_DEBUG", "").lower() in ("true", "1", "t"):
            self.logger.setLevel(logging.DEBUG)

    def log(self, msg, level="info"):
        if level not in _COLORS:
            raise ValueError(f"Unknown log level: {level}")
        text = f"{_COLORS[level]}> [{datetime.now().strftime('%Y-%m-%d %H:%M:%S')}] {msg}\033[0m"
        {"info": self.logger.info, "success": self.logger.info, "warning": self.logger.warning,
         "error": self.logger.error, "debug": self.logger.debug}[level](text)


logger = MyLogger()


def seed_everything(seed: int):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


@torch.inference_mode()
def sample_token(logits: torch.Tensor, temperature=0.6, top_p=0.95, top_k=-1):
    """logits [bsz, vocab] -> token ids [bsz, 1].  temperature 0 = argmax; otherwise nucleus (top-p)
    sampling, optionally restricted to the top-k logits (the reference's AMD path only has argmax)."""
    if temperature == 0.0:
        return logits.argmax(dim=-1, keepdim=True)
    logits = logits.float() / temperature
    if top_k and top_k > 0:
        kth = torch.topk(logits, min(top_k, logits.shape[-1]), dim=-1).values[..., -1:]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    probs = logits.softmax(dim=-1)
    sp, si = torch.sort(probs, dim=-1, descending=True)
    keep = (sp.cumsum(-1) - sp) < top_p          # always keeps the most likely token
    sp = sp * keep
    sp = sp / sp.sum(-1, keepdim=True)
    pick = torch.multinomial(sp, 1)
    return si.gather(-1, pick)


@torch.no_grad()
def init_model_cpu(model_name: str, dtype: torch.dtype, arch=None, device=None):
    """Weight source for ``model_name`` (the name is the reference's; the random source builds its
    tensors directly on ``device``)."""
    from .config import ModelConfig, load_arch_config

    arch = arch or load_arch_config(ModelConfig(model_name=model_name))
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu"
    return load_weight_source(model_name, arch, dtype, device)
