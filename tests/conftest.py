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


@pytest.fixture(scope="session", autouse=True)
def _generator_graph_state_outside_inference_mode():
    """torch keeps the default generator's seed / offset tensors for graph capture only while some captured graph is
    alive, and creates them again at the next capture - as INFERENCE tensors when that capture runs under
    ``torch.inference_mode()`` (the engine and the mega back-end capture there).  A later capture outside inference
    mode then fails in ``capture_begin`` ("Inplace update to inference tensor") for as long as the earlier graph has not
    been collected, which made the suite depend on garbage-collection timing.  One small graph captured outside
    inference mode and kept for the session pins normal tensors."""
    try:
        import torch

        if not torch.cuda.is_available():
            yield None
            return
    except Exception:
        yield None
        return
    buf = torch.zeros(8, device="cuda")
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        buf.add_(1)
    torch.cuda.current_stream().wait_stream(stream)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        buf.add_(1)
    torch.cuda.synchronize()
    yield graph
    del graph
