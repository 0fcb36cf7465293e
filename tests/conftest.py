"""pytest configuration.

* ``gpu`` marker: tests that need an MI355X.  They are skipped automatically when no GPU is visible,
  so ``pytest -m "not gpu"`` (CPU container) and ``pytest -m gpu`` (GPU box) both work.
* Multi-rank kernels are tested with virtual ranks inside this one process; a rank that spins on a
  peer must own a hardware queue of its own, so ask HIP for more queues than its default of 4 before
  the runtime initialises.
"""
import os
import sys
from pathlib import Path

os.environ["GPU_MAX_HW_QUEUES"] = "24"  # the box exports 4; virtual ranks need one queue per stream
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import pytest  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU")


def pytest_collection_modifyitems(config, items):
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
