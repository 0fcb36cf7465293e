"""Tensor-parallel gated MLP (Qwen / Llama style).

Reference: layers/amd/tp_mlp.py:34-212 (shard_local, TP_MLP) plus the all-reduce and GEMM+AR
forward modes of layers/nvidia/tp_mlp.py:172-228.  gate/up are column-parallel and fused into one
[2*I/w, H] weight, down is row-parallel [H, I/w].

Forward modes
* ``torch_fwd``               - plain GEMMs + RCCL all-reduce (baseline / golden)
* ``dist_triton_fwd``         - AG+GEMM -> fused SiLU*up kernel -> GEMM+RS (activations stay M-sharded)
* ``dist_triton_AR_fwd``      - local GEMMs + symmetric-memory all-reduce kernel
* ``dist_triton_gemm_ar_fwd`` - second GEMM fused with the all-reduce
"""
from __future__ import annotations

import torch

from ..ops import elementwise
from ..ops.ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context
from ..ops.allreduce import AllReduceMethod, all_reduce, create_allreduce_ctx
from ..ops.gemm import gemm_nt, gemm_nt_norm, gemm_nt_silu_mul, norm_gemm_supported
from ..ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node
from ..utils.distributed import get_env


def shard_local(tensor: torch.Tensor, world_size: int, dim: int, local_rank: int) -> torch.Tensor:
    """This rank's equal share of ``tensor`` along ``dim`` (contiguous copy)."""
    if not 0 <= dim < tensor.dim():
        raise ValueError(f"shard_local: dim={dim} but the tensor has shape {tuple(tensor.shape)}")
    if not 0 <= local_rank < world_size:
        raise ValueError(f"shard_local: rank {local_rank} outside [0, {world_size})")
    n = tensor.shape[dim]
    per, rem = divmod(n, world_size)
    if rem != 0 or per == 0:
        raise ValueError(f"shard_local: cannot cut {n} entries of dim {dim} into {world_size} equal non-empty shards")
    return tensor.narrow(dim, local_rank * per, per).contiguous()


def prenorm_generic(layer, x, residual, norm_w, eps, *args):
    """RMSNorm of ``x`` (+ ``residual``) as its own kernel, then ``layer.fwd``: -> (block output, x + residual)."""
    if residual is None:
        return layer.fwd(elementwise.rms_norm(x, norm_w, eps), *args), x
    y, hidden = elementwise.rms_norm(x.reshape(residual.shape), norm_w, eps, residual=residual)
    return layer.fwd(y, *args), hidden


def _linear(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x [..., K] @ w[N, K]^T through the native GEMM when the shape allows it."""
    x2 = x.reshape(-1, x.shape[-1])
    return gemm_nt(x2, w).view(*x.shape[:-1], w.shape[0])


class TP_MLP:

    def __init__(self, rank=0, world_size=8, group=None, env=None):
        self.rank = rank
        self.world_size = world_size
        self.group = group
        self.env = env
        self.act_fn = None
        self.gate_up_proj = None
        self.down_proj = None
        self.ag_ctx = self.rs_ctx = self.ar_ctx = self.gemm_ar_ctx = None
        self.ar_method = AllReduceMethod.Unknown

    def _env(self):
        return self.env or get_env()

    def _device(self):
        try:
            return self._env().device
        except RuntimeError:
            return torch.device("cuda" if torch.cuda.is_available() else "cpu")

    def _init_parameters(self, mlp, verbose=False):
        """``mlp`` is any object with gate_proj / up_proj / down_proj ``.weight`` (HF module or the
        built-in random-init tree)."""
        dev = self._device()
        gate = shard_local(mlp.gate_proj.weight.detach(), self.world_size, 0, self.rank)
        up = shard_local(mlp.up_proj.weight.detach(), self.world_size, 0, self.rank)
        self.gate_up_proj = torch.cat((gate, up), dim=0).to(dev)          # [2 * I / w, H]
        self.down_proj = shard_local(mlp.down_proj.weight.detach(), self.world_size, 1, self.rank).to(dev)  # [H, I / w]
        self.act_fn = getattr(mlp, "act_fn", torch.nn.functional.silu)
        self.ag_N_per_rank = self.gate_up_proj.shape[0]
        self.K = self.gate_up_proj.shape[1]
        self.dtype = self.gate_up_proj.dtype
        assert getattr(mlp.gate_proj, "bias", None) is None, "bias is not supported"
        if verbose:
            print(f"[RANK {self.rank}] MLP gate_up {tuple(self.gate_up_proj.shape)} down {tuple(self.down_proj.shape)}")

    def _init_ctx(self, max_M, ag_intranode_stream=None, BLOCK_M=256, BLOCK_N=256, BLOCK_K=64, stages=2, serial=False,
                  ag_internode_stream=None):
        self.ag_ctx = create_ag_gemm_intra_node_context(
            max_M=max_M, N=self.ag_N_per_rank, K=self.K, rank=self.rank, num_ranks=self.world_size,
            input_dtype=self.dtype, output_dtype=self.dtype, tp_group=self.group, ag_streams=ag_intranode_stream,
            serial=serial, autotune=True, BLOCK_M=BLOCK_M, BLOCK_N=BLOCK_N, BLOCK_K=BLOCK_K, stages=stages,
            M_PER_CHUNK=256, env=self.env)
        self.rs_ctx = create_gemm_rs_intra_node_context(max_M=max_M, N=self.K, rank=self.rank,
                                                        num_ranks=self.world_size, output_dtype=self.dtype,
                                                        tp_group=self.group, fuse_scatter=True, env=self.env)

    def _init_AR_ctx(self, max_M, method: AllReduceMethod = AllReduceMethod.Unknown, dtype=torch.bfloat16):
        self.ar_method = method
        N = self.down_proj.shape[0]
        self.ar_ctx = create_allreduce_ctx(workspace_nbytes=max(max_M * N * dtype.itemsize, 1 << 16) * 2, rank=self.rank,
                                           world_size=self.world_size, local_world_size=self.world_size, env=self.env)

    def _init_gemm_ar_ctx(self, max_M, dtype=torch.bfloat16):
        from .gemm_allreduce_layer import GemmARLayer

        N, K = self.down_proj.shape
        self.gemm_ar_ctx = GemmARLayer(self.group, max_M, N, K, dtype, dtype, self.world_size, persistent=True,
                                       use_ll_kernel=max_M <= 256, copy_to_local=False, rank=self.rank, env=self.env)

    def finalize(self):
        self.ag_ctx = self.rs_ctx = self.ar_ctx = self.gemm_ar_ctx = None

    # ---- shared pieces ------------------------------------------------------------------------
    def _is_silu(self) -> bool:
        return self.act_fn in (torch.nn.functional.silu, None) or getattr(self.act_fn, "__name__", "") == "silu" \
            or type(self.act_fn).__name__ in ("SiLU", "SiLUActivation")

    def _gate_up_act(self, x: torch.Tensor) -> torch.Tensor:
        """act(x @ gate^T) * (x @ up^T); decode-sized inputs take one launch (GEMM with the activation
        in its epilogue)."""
        x2 = x.reshape(-1, x.shape[-1])
        if x2.shape[0] <= 16 and x2.is_cuda and self._is_silu():
            return gemm_nt_silu_mul(x2, self.gate_up_proj).view(*x.shape[:-1], -1)
        return self._act(_linear(x, self.gate_up_proj))

    def _act(self, gate_up: torch.Tensor) -> torch.Tensor:
        if self._is_silu():
            return elementwise.silu_mul(gate_up)
        g, u = torch.chunk(gate_up, 2, dim=-1)
        return self.act_fn(g) * u

    def _all_reduce_torch(self, out):
        if self.world_size > 1 and torch.distributed.is_initialized():
            torch.distributed.all_reduce(out, torch.distributed.ReduceOp.SUM, group=self.group)
        return out

    # ---- forward modes ------------------------------------------------------------------------
    @torch.inference_mode()
    def torch_fwd(self, x):
        out = torch.nn.functional.linear(x, self.gate_up_proj)
        g, u = torch.chunk(out, 2, dim=-1)
        out = torch.nn.functional.linear(torch.nn.functional.silu(g) * u if self.act_fn is None else self.act_fn(g) * u,
                                         self.down_proj)
        return self._all_reduce_torch(out)

    @torch.inference_mode()
    def dist_triton_fwd(self, x):
        """x [M/w, H] or [B/w, S, H] (row-sharded) -> same shape."""
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        out = ag_gemm_intra_node(x2, self.gate_up_proj, transe_b=False, ctx=self.ag_ctx)
        out = self._act(out)
        out = gemm_rs_intra_node(out, self.down_proj, self.rs_ctx)
        return out.view(*shape[:-1], -1)

    @torch.inference_mode()
    def dist_triton_AR_fwd(self, x, act=None):
        """x replicated [M, H]; GEMMs are local, the sum runs on the symmetric-memory all-reduce.  ``act``: the
        activated gate / up product when the caller has computed it already (``fwd_prenorm``)."""
        out = self._gate_up_act(x) if act is None else act
        out = _linear(out, self.down_proj)
        if self.world_size > 1:
            assert self.ar_ctx is not None, "AllReduce context is not initialized."
            out = all_reduce(out.contiguous(), output=torch.empty_like(out), method=self.ar_method, ctx=self.ar_ctx)
        return out.view_as(x)

    @torch.inference_mode()
    def dist_triton_gemm_ar_fwd(self, x, act=None):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        assert self.gemm_ar_ctx is not None, "GemmAR context is not initialized."
        out = self._act(_linear(x2, self.gate_up_proj)) if act is None else act.reshape(x2.shape[0], -1)
        out = self.gemm_ar_ctx.forward(out, self.down_proj)
        return out.view(*shape[:-1], -1)

    @torch.inference_mode()
    def fwd(self, x: torch.Tensor):
        raise NotImplementedError("Please use torch_fwd or dist_triton_fwd instead.")

    @torch.inference_mode()
    def fwd_prenorm(self, x, residual, norm_w, eps):
        """RMSNorm(x [+ residual]) -> this block: (block output, x + residual).  Decode-sized inputs of the modes
        whose first projection is a local GEMM run the norm inside that GEMM's launch (``gemm_nt_norm``)."""
        fn = getattr(self.fwd, "__func__", None)
        solo = fn is TP_MLP.dist_triton_fwd and self.world_size == 1      # one rank: local GEMMs, like the all-reduce form
        if (fn in (TP_MLP.dist_triton_AR_fwd, TP_MLP.dist_triton_gemm_ar_fwd) or solo) and self._is_silu():
            x2 = x.reshape(-1, x.shape[-1])
            r2 = residual.reshape(x2.shape) if residual is not None else None
            if norm_gemm_supported(x2, self.gate_up_proj, norm_w, r2):
                act, hidden = gemm_nt_norm(x2, norm_w, eps, self.gate_up_proj, residual=r2, silu=True)
                shape = residual.shape if residual is not None else x.shape
                fwd = self.dist_triton_AR_fwd if solo else self.fwd
                return fwd(x.reshape(shape), act=act.view(*shape[:-1], -1)), hidden.view(shape)
        return prenorm_generic(self, x, residual, norm_w, eps)

    # ---- single-op helpers (used by the per-op tests / benchmarks) --------------------------------
    @torch.inference_mode()
    def torch_ag_gemm(self, x: torch.Tensor):
        M_per_rank, K = x.shape
        if self.world_size == 1 or not torch.distributed.is_initialized():
            return torch.matmul(x, self.gate_up_proj.T)
        if not hasattr(self, "ag_buffer") or self.ag_buffer.shape[0] != M_per_rank * self.world_size:
            self.ag_buffer = torch.empty([M_per_rank * self.world_size, K], dtype=x.dtype, device=x.device)
        torch.distributed.all_gather_into_tensor(self.ag_buffer, x, group=self.group)
        return torch.matmul(self.ag_buffer, self.gate_up_proj.T)

    @torch.inference_mode()
    def dist_triton_ag_gemm(self, x: torch.Tensor):
        return ag_gemm_intra_node(x, self.gate_up_proj, transe_b=False, ctx=self.ag_ctx)

    @torch.inference_mode()
    def torch_gemm_rs(self, x: torch.Tensor):
        gemm_out = torch.matmul(x, self.down_proj.T)
        if self.world_size == 1 or not torch.distributed.is_initialized():
            return gemm_out
        M = x.shape[0]
        if not hasattr(self, "rs_buffer") or self.rs_buffer.shape[0] != M // self.world_size:
            self.rs_buffer = torch.empty([M // self.world_size, self.down_proj.shape[0]], dtype=x.dtype, device=x.device)
        torch.distributed.reduce_scatter_tensor(self.rs_buffer, gemm_out, group=self.group)
        return self.rs_buffer

    @torch.inference_mode()
    def dist_triton_gemm_rs(self, x: torch.Tensor):
        return gemm_rs_intra_node(x, self.down_proj, self.rs_ctx)
