"""Import-compatible facade: the reference's ``triton_dist`` module tree mapped onto the
MI355X-native implementation in ``triton_distributed_gh_amd`` (reference: python/triton_dist/__init__.py).

Sub-modules mirror the reference's import paths (``triton_dist.utils``, ``triton_dist.kernels.amd``,
``triton_dist.kernels.nvidia``, ``triton_dist.layers``, ``triton_dist.models``, ``triton_dist.language``,
``triton_dist.tools``, ``triton_dist.mega_triton_kernel``); every name resolves to a HIP/C++-backed
implementation - there is no Triton JIT underneath.
"""
import triton_distributed_gh_amd as _impl

__version__ = _impl.__version__ if hasattr(_impl, "__version__") else "0.1.0"
