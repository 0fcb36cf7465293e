"""Tensor-parallel Mixture-of-Experts block (every expert's FFN is sharded over the TP ranks).

Reference: layers/nvidia/tp_moe.py:42-279 (the reference has no AMD implementation and its Qwen3
MoE model raises NotImplemented on AMD, models/qwen_moe.py:164-165).

Pipeline of ``dist_triton_fwd``: router softmax / top-k on the local tokens -> all-gather of the
(ids, weights) pairs with the low-latency symmetric-memory all-gather (the reference uses two NCCL
all_gathers here) -> AG + grouped GEMM (gate_up) -> fused SiLU*up -> grouped GEMM with routing
weights + top-k reduce + reduce-scatter.

Weights are kept K-contiguous ([E, N, K], the layout the MFMA grouped GEMM reads); the reference
stores the transposes.  The per-rank intermediate size is zero-padded to a multiple of 64 (>= 128)
so the down projection's K meets the GEMM tile constraints (e.g. 768 / 8 = 96 -> 128).
"""
from __future__ import annotations

import torch

from ..ops import elementwise
from ..ops.allgather import create_fast_allgather_context, fast_allgather
from ..ops.allreduce import all_reduce, create_allreduce_ctx
from ..ops.gemm import gemm_nt, gemm_supported
from ..ops.moe import (MOE_GROUP_MID, MOE_GROUP_MID_ROWS_PER_BIG_EXPERT, MOE_GROUP_MID_ROWS_PER_EXPERT, ag_group_gemm, create_ag_group_gemm_context, create_moe_rs_context, MOE_DECODE_MAX_TOKENS, MOE_GROUP16_MAX_TOKENS, group16_row_limit, moe_decode_fwd,
                       moe_decode_supported, moe_group_mid_fwd, moe_group_mid_preferred, moe_group_mid_supported, moe_route,
                       moe_route_supported,
                       run_moe_reduce_rs)
from ..utils.distributed import get_env

# most tokens (over all ranks) the gather -> all experts' local slices -> all-reduce form of the layer takes
MOE_DECODE_TP_MAX_TOKENS = 1024
from .tp_mlp import shard_local


def _pad_to(n: int) -> int:
    return max(128, (n + 63) // 64 * 64)


class TP_MoE:

    def __init__(self, rank=0, world_size=8, group=None, autotune=False, env=None):
        self.rank = rank
        self.world_size = world_size
        self.group = group
        self.env = env
        self.act_fn = None
        self.gate_up_proj = None     # [E, 2 * Ip, H]   (Ip = padded I / w)
        self.down_proj = None        # [E, H, Ip]
        self.ag_ctx = self.rs_ctx = None
        self.top_k = self.num_experts = self.gate = None
        self.norm_topk_prob = True
        self.ag_group_gemm = ag_group_gemm
        self.run_moe_reduce_rs = run_moe_reduce_rs

    def _env(self):
        return self.env or get_env()

    def _device(self):
        try:
            return self._env().device
        except RuntimeError:
            return torch.device("cuda" if torch.cuda.is_available() else "cpu")

    def _init_parameters(self, mlp, verbose=False):
        """``mlp`` has ``gate.weight [E, H]``, ``experts[e].{gate,up,down}_proj.weight``, ``num_experts``
        and ``top_k``."""
        dev = self._device()
        w = self.world_size
        self.num_experts = int(mlp.num_experts)
        self.top_k = int(mlp.top_k)
        self.norm_topk_prob = bool(getattr(mlp, "norm_topk_prob", True))
        self.gate = mlp.gate.weight.detach().to(dev)
        H = self.gate.shape[1]
        self.hidden_size = H
        I = mlp.experts[0].gate_proj.weight.shape[0]
        dtype = mlp.experts[0].gate_proj.weight.dtype
        Il = I // w
        Ip = _pad_to(Il)
        self.inter_local, self.inter_padded = Il, Ip
        self.gate_up_proj = torch.zeros(self.num_experts, 2 * Ip, H, dtype=dtype, device=dev)
        self.down_proj = torch.zeros(self.num_experts, H, Ip, dtype=dtype, device=dev)
        for e in range(self.num_experts):
            ex = mlp.experts[e]
            self.gate_up_proj[e, :Il] = shard_local(ex.gate_proj.weight.detach(), w, 0, self.rank).to(dev)
            self.gate_up_proj[e, Ip:Ip + Il] = shard_local(ex.up_proj.weight.detach(), w, 0, self.rank).to(dev)
            self.down_proj[e, :, :Il] = shard_local(ex.down_proj.weight.detach(), w, 1, self.rank).to(dev)
        self.act_fn = getattr(mlp.experts[0], "act_fn", torch.nn.functional.silu)
        self.dtype = dtype
        if verbose:
            print(f"[RANK {self.rank}] MoE gate_up {tuple(self.gate_up_proj.shape)} down {tuple(self.down_proj.shape)}")

    def _init_ctx(self, M, ag_streams=None):
        """``M`` = max tokens over ALL ranks per forward."""
        env = self._env()
        w = self.world_size
        self.max_M = M
        self.ag_ctx = create_ag_group_gemm_context(M, self.gate_up_proj.shape[1], self.hidden_size, self.num_experts,
                                                   self.top_k, self.dtype, self.rank, w, w, ag_streams=ag_streams,
                                                   env=self.env)
        self.rs_ctx = create_moe_rs_context(rank=self.rank, world_size=w, local_world_size=w, max_token_num=M,
                                            hidden_dim=self.hidden_size, num_experts=self.num_experts, topk=self.top_k,
                                            input_dtype=self.dtype, env=self.env)
        # router metadata all-gather: per token topk int32 ids + topk fp32 weights
        self.meta_bytes = (M // w) * self.top_k * 8
        self.meta_ctx = create_fast_allgather_context(self.rank, max_buffer_size=self.meta_bytes, env=self.env)
        self.meta_buf = env.heap.create_tensor([2, w, self.meta_bytes // 4], torch.int32)
        self.meta_state = {"calls": 0}     # shared by every layer that shares this context
        # decode path on several ranks (``_decode_fwd_tp``): the tokens' rows are all-gathered (low-latency form),
        # every rank runs ALL tokens against its slice of the experts' intermediate dimension, partial sums are
        # all-reduced
        self.dec_per = max(1, min(M, MOE_DECODE_TP_MAX_TOKENS) // w)
        self.dec_x_ctx = self.dec_x_buf = self.dec_ar_ctx = None
        if w > 1 and env.backend == "hip":
            xb = self.dec_per * self.hidden_size * self.dtype.itemsize
            self.dec_x_ctx = create_fast_allgather_context(self.rank, max_buffer_size=xb, env=self.env)
            self.dec_x_buf = env.heap.create_tensor([2, w, xb // 4], torch.int32)
            self.dec_ar_ctx = create_allreduce_ctx(workspace_nbytes=max(self.dec_per * w * self.hidden_size * self.dtype.itemsize,
                                                                        1 << 16) * 2,
                                                   rank=self.rank, world_size=w, local_world_size=w, env=self.env)

    def finalize(self):
        self.ag_ctx = self.rs_ctx = None

    def _route(self, x2: torch.Tensor, fused: bool = False):
        if fused and x2.shape[0] <= 16 and gemm_supported(x2, self.gate):
            logits = gemm_nt(x2, self.gate)          # decode: the weight-streaming kernel (3 us at one row, library 8.5)
        else:
            logits = torch.nn.functional.linear(x2, self.gate)
        if fused and moe_route_supported(logits, self.top_k):
            return moe_route(logits, self.top_k, self.norm_topk_prob)      # one launch (ops/moe.py)
        probs = torch.nn.functional.softmax(logits, dim=1, dtype=torch.float)
        wts, ids = torch.topk(probs, self.top_k, dim=-1)
        if self.norm_topk_prob:
            wts = wts / wts.sum(dim=-1, keepdim=True)
        return wts, ids

    def _act(self, gate_up):
        return elementwise.silu_mul(gate_up)

    @torch.inference_mode()
    def torch_fwd(self, x):
        """Golden: loop over experts, x replicated [bsz, seq, H]; output all-reduced.  A call made while a HIP graph
        is being captured, where the loop's ``torch.where`` would have to synchronise, takes ``torch_fwd_no_loop``
        (same arithmetic up to the order of the top-k sum; a randomly initialised MoE model amplifies even that
        through flipped routing decisions, so eager goldens stay on the loop)."""
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if x.is_cuda and torch.cuda.is_current_stream_capturing():
            return self.torch_fwd_no_loop(x)
        wts, ids = self._route(x2)
        wts = wts.to(x.dtype)
        Ip = self.inter_padded
        out = torch.zeros_like(x2)
        for e in range(self.num_experts):
            tok, k = torch.where(ids == e)
            if tok.numel() == 0:
                continue
            h = x2[tok] @ self.gate_up_proj[e].t()
            h = torch.nn.functional.silu(h[:, :Ip]) * h[:, Ip:]
            h = (h @ self.down_proj[e].t()) * wts[tok, k, None]
            out.index_add_(0, tok, h)
        if self.world_size > 1 and torch.distributed.is_initialized():
            torch.distributed.all_reduce(out, torch.distributed.ReduceOp.SUM, group=self.group)
        return out.view(shape)

    @torch.inference_mode()
    def torch_fwd_no_loop(self, x):
        """The same result without a loop over experts and without a host synchronisation (reference:
        layers/nvidia/tp_moe.py torch_fwd_no_loop): the weights of every (token, expert) pair are gathered and
        multiplied as one batch, 3 x I x H x 2 bytes per pair, so this is for decode-sized inputs."""
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        wts, ids = self._route(x2)
        n, k = ids.shape
        Ip = self.inter_padded
        flat = ids.reshape(-1)
        xk = x2.repeat_interleave(k, 0).unsqueeze(2)                       # [n k, H, 1]
        h = torch.bmm(self.gate_up_proj[flat], xk).squeeze(2)               # [n k, 2 Ip]
        h = torch.nn.functional.silu(h[:, :Ip]) * h[:, Ip:]
        o = torch.bmm(self.down_proj[flat], h.unsqueeze(2)).squeeze(2)      # [n k, H]
        o = o.view(n, k, -1) * wts.to(x.dtype).unsqueeze(-1)                # rounded like the loop's products
        out = o[:, 0]
        for j in range(1, k):
            out = out + o[:, j]
        if self.world_size > 1 and torch.distributed.is_initialized():
            torch.distributed.all_reduce(out, torch.distributed.ReduceOp.SUM, group=self.group)
        return out.view(shape)

    def _gather_meta(self, wts: torch.Tensor, ids: torch.Tensor):
        w = self.world_size
        n, k = ids.shape
        if w == 1:
            return wts.contiguous(), ids.to(torch.int32).contiguous()
        assert 2 * n * k * 4 <= self.meta_bytes, "more tokens than the context was sized for"
        full = self.meta_buf[self.meta_state["calls"] & 1]   # [w, words]; slots are only written locally
        self.meta_state["calls"] += 1
        mine = full[self.rank]
        mine[: n * k].copy_(ids.reshape(-1).to(torch.int32))
        mine[n * k: 2 * n * k].copy_(wts.reshape(-1).float().contiguous().view(torch.int32))
        fast_allgather(full, self.meta_ctx, mode="push_2d_ll")
        ids_full = full[:, : n * k].reshape(w * n, k)
        wts_full = full[:, n * k: 2 * n * k].view(torch.float32).reshape(w * n, k)
        return wts_full, ids_full

    def _decode_fwd_tp(self, x2: torch.Tensor):
        """At most ``MOE_DECODE_MAX_TOKENS`` tokens over all ranks, ``x2`` = this rank's rows: gather the rows, route
        and run every token here against the local slice of the experts (``moe_decode_fwd``), all-reduce, keep the
        own rows.  Every rank computes the same routing from the same gathered rows."""
        w, n_local, H = self.world_size, x2.shape[0], x2.shape[1]
        words = n_local * H * x2.element_size() // 4
        full = self.dec_x_buf[self.meta_state["calls"] & 1]       # [w, words of dec_per rows]; parity as for meta_buf
        self.meta_state["calls"] += 1
        full[self.rank][:words].copy_(x2.view(torch.int32).reshape(-1))
        fast_allgather(full, self.dec_x_ctx, mode="push_2d_ll")
        xall = full[:, :words].reshape(-1).view(x2.dtype).view(w * n_local, H)
        wts, ids = self._route(xall, fused=True)
        E, two_ip, _ = self.gate_up_proj.shape
        if (moe_group_mid_supported(xall, self.gate_up_proj, self.down_proj, self.top_k)
                and moe_group_mid_preferred(xall.shape[0], self.top_k, E, H, two_ip // 2)):
            part = moe_group_mid_fwd(xall, ids, wts, self.gate_up_proj, self.down_proj)
        else:
            part = moe_decode_fwd(xall, ids, wts, self.gate_up_proj, self.down_proj)
        tot = all_reduce(part, ctx=self.dec_ar_ctx, output=torch.empty_like(part))
        return tot[self.rank * n_local:(self.rank + 1) * n_local]

    def _decode_tp_ok(self, x2: torch.Tensor) -> bool:
        """``moe_decode_supported`` for the gathered batch (``world_size`` times this rank's rows)."""
        n_all = x2.shape[0] * self.world_size
        E, two_ip, H = self.gate_up_proj.shape
        if not moe_decode_supported(x2[:1], self.gate_up_proj, self.down_proj, self.top_k):
            return False
        if n_all <= MOE_DECODE_MAX_TOKENS or (n_all <= MOE_GROUP16_MAX_TOKENS and
                                              n_all * self.top_k <= group16_row_limit(E, H, two_ip // 2) and E <= 1024
                                              and two_ip // 2 >= 64):
            return True
        # (the 64- / 128-row form: same gather / all-reduce around it)
        Ip = two_ip // 2
        big = 3 * H * Ip * 2 > (32 << 20)
        limit = (MOE_GROUP_MID_ROWS_PER_BIG_EXPERT if big else MOE_GROUP_MID_ROWS_PER_EXPERT) * E
        return (MOE_GROUP_MID and n_all <= MOE_DECODE_TP_MAX_TOKENS and H % 64 == 0 and Ip % 64 == 0 and Ip >= 128
                and E <= 1024 and n_all * self.top_k <= limit)

    @torch.inference_mode()
    def dist_triton_fwd(self, x: torch.Tensor):
        """x token-sharded [bsz / w, seq, H] (or [tokens / w, H]) -> same shape."""
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        if (self.world_size > 1 and getattr(self, "dec_x_buf", None) is not None and x2.shape[0] <= self.dec_per
                and x2.shape[1] % 2 == 0 and self._decode_tp_ok(x2)):
            return self._decode_fwd_tp(x2).reshape(shape)
        wts, ids = self._route(x2, fused=True)
        if self.world_size == 1:
            E, two_ip, H = self.gate_up_proj.shape
            mid = (moe_group_mid_supported(x2, self.gate_up_proj, self.down_proj, self.top_k)
                   and moe_group_mid_preferred(x2.shape[0], self.top_k, E, H, two_ip // 2))
            if not mid and moe_decode_supported(x2, self.gate_up_proj, self.down_proj, self.top_k):
                # a decode step: weight-streaming launches over the tokens' own experts, per (token, expert) pair or per
                # sixteen-row block of an expert (ops/moe.py: moe_decode_fwd)
                return moe_decode_fwd(x2, ids, wts, self.gate_up_proj, self.down_proj).view(shape)
            if mid:
                # tens to a few hundred rows per expert: blocks of 64 / 128 rows on the mid-M GEMM body
                return moe_group_mid_fwd(x2, ids, wts, self.gate_up_proj, self.down_proj).view(shape)
        full_w, full_ids = self._gather_meta(wts, ids)
        h = self.ag_group_gemm(x2, self.gate_up_proj, ctx=self.ag_ctx, full_topk_ids=full_ids.contiguous())
        h = self._act(h)
        # one rank: the sort by expert that ag_group_gemm just made (world size 1, rank 0) is the one the second
        # grouped GEMM needs
        extra = {"idx": getattr(self.ag_ctx, "last_index", None)} if self.world_size == 1 else {}
        out = self.run_moe_reduce_rs(h, self.down_proj, full_ids.contiguous(), full_w.contiguous(), ctx=self.rs_ctx, **extra)
        return out.view(shape)

    def fwd(self, x: torch.Tensor):
        raise NotImplementedError("Please use torch_fwd or dist_triton_fwd instead.")
