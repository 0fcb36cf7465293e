"""Small-message all-gather layer (reference: layers/nvidia/low_latency_allgather_layer.py:30-187).

Keeps the reference's method names.  One MI355X node is a full xGMI mesh, so the 2-D / 3-D / NUMA
ring variants all resolve to the full-mesh push (or its flag-in-data "LL" form) and ``pull`` to
the full-mesh pull; ``signal_target`` increases monotonically and doubles as the LL flag value."""
from __future__ import annotations

import torch

from ..ops.allgather import create_fast_allgather_context, fast_allgather
from ..utils.distributed import get_env


class AllGatherLayer(torch.nn.Module):

    def __init__(self, nnodes, world_size, rank, max_buffer_size: int = 2 * 32 * 128 * 128, stages=2, env=None):
        super().__init__()
        self.env = env or get_env()
        self.nnodes, self.world_size, self.rank = nnodes, world_size, rank
        self.stages = stages
        self.max_buffer_size = max_buffer_size
        self.ctx = create_fast_allgather_context(rank, 0, world_size, nnodes, max_buffer_size=max_buffer_size,
                                                 env=self.env)

    @property
    def signal_target(self) -> int:
        return self.ctx.signal_target + 1

    def finalize(self):
        self.ctx = None

    def _run(self, symm_buffer: torch.Tensor, mode: str) -> torch.Tensor:
        w = self.world_size
        buf = symm_buffer if symm_buffer.shape[0] == w else symm_buffer.view(w, -1)
        fast_allgather(buf, self.ctx, mode=mode)
        return symm_buffer

    def forward_pull(self, symm_buffer):
        return self._run(symm_buffer, "pull")

    def forward_push_2d(self, symm_buffer):
        return self._run(symm_buffer, "push_2d")

    def forward_push_3d(self, symm_buffer):
        return self._run(symm_buffer, "push_2d")

    def forward_push_3d_ll(self, symm_buffer):
        return self._run(symm_buffer, "push_2d_ll")

    def forward_push_2d_ll(self, symm_buffer):
        return self._run(symm_buffer, "push_2d_ll")

    def forward_push_numa_2d(self, symm_buffer):
        return self._run(symm_buffer, "push_2d")

    def forward_push_numa_2d_ll(self, symm_buffer):
        return self._run(symm_buffer, "push_2d_ll")

    def forward_push_numa_2d_ll_multinode(self, symm_buffer):
        return self._run(symm_buffer, "push_2d_ll")

    def forward_push_2d_ll_multimem(self, symm_buffer):
        return self._run(symm_buffer, "push_2d_ll")
