"""MI355X-native compute/communication-overlap framework (core package).

Layout
  csrc/      HIP device headers, gfx950 kernels, C++ runtime (built in-tree by ``build.py``)
  utils/     distributed bootstrap, symmetric heap, perf / profiling / topology helpers
  ops/       Python entry points of every kernel and fused op
  parallel/  TP / EP / SP / PP layers built from the ops
  models/    Qwen3 dense / MoE models, KV cache, inference engine
  mega/      task-graph -> single persistent kernel runtime
  tools/     AOT kernel registry and CLI

The reference-compatible import paths (``triton_dist.utils``, ``triton_dist.kernels.amd`` ...) are a
thin facade over this package.
"""
__version__ = "0.1.0"
