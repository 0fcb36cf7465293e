"""Second back-end of ``ModelBuilder``: the recorded op list lowered onto the library's fused launches and captured
into ONE HIP graph.

The interpreter back-end (csrc/ops/mega_kernel.hip) runs the task graph as one persistent kernel; on this chip a
dependent hop through its scoreboard costs 3.6 us against 1.2 - 1.5 us for a kernel boundary inside a replayed
graph, and the weight-streaming stages run at the same speed in both, so the graph of fused kernels is the faster
way to execute the same graph (reference: python/triton_dist/mega_triton_kernel/models/model_builder.py:372-404
``compile`` / ``run``; its result - mega 3.33 ms vs CUDA graph 4.65 ms on H800,
docs/mega_triton_kernel.md:32 - rests on a boundary cost that HIP graphs on MI355X do not have).

Lowering walks the ops in recorded order (a topological order) and fuses, where the shapes allow:

    rms_norm -> linear                    ->  one launch (the norm is folded into the weight-streaming GEMM;
                                              mode 1: with the SiLU * up epilogue as well)
    qk_norm_rope_update_kvcache -> attn   ->  the attention kernel norms, rotates and appends its own heads
                                              (+ the head-parallel merge launch)
    linear (mode 1)                       ->  decode GEMM with SiLU * up in the epilogue
    linear -> allreduce, rms_norm, add, silu_mul, prefetch (dropped: nothing to prefetch for between launches)

A fused launch returns fresh tensors instead of writing the recorded buffers; ``vals`` maps a recorded tensor's
address to the tensor that currently holds its value (the same tensor on every replay, since the steps are captured).
At the end of a step the values are copied back into the recorded tensors, so callers see results exactly where
the interpreter leaves them: all of them by default, or only the tensors named in ``outputs`` (a model that reads
nothing but its logits saves the copies of the intermediate buffers).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch

from .core import OpRecord, TaskType


class LoweringError(NotImplementedError):
    pass


class GraphProgram:
    def __init__(self, builder, outputs=None):
        self.b = builder
        self._wanted = None if outputs is None else {t.data_ptr() for t in outputs}
        self.steps: List[Callable[[Dict[int, torch.Tensor]], None]] = []
        self.launch_names: List[str] = []
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._recorded: Dict[int, torch.Tensor] = {}
        self._ar_ctx = None
        self._total_scratch: Dict[int, torch.Tensor] = {}
        self._pos64: Dict[int, tuple] = {}        # kv_lens address -> (kv_lens, int64 copy made once per step)
        self._seq_tables: Dict[int, torch.Tensor] = {}
        self.calls = 0
        self._capture_stream = None
        self._lower()

    # ---- helpers --------------------------------------------------------------------------------------------
    @staticmethod
    def _get(vals, t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        if t is None:
            return None
        v = vals.get(t.data_ptr())
        return t if v is None else v.view(t.shape)

    def _readers_until_rewrite(self, ptr: int, after: int) -> List[int]:
        out = []
        for op in self.b.ops[after + 1:]:
            if ptr in op.reads:
                out.append(op.op_id)
            if ptr in op.writes:
                break
        return out

    def _emit(self, name: str, fn) -> None:
        self.launch_names.append(name)
        self.steps.append(fn)

    # ---- lowering ---------------------------------------------------------------------------------------------
    def _lower(self) -> None:
        ops = [op for op in self.b.ops if op.type != TaskType.PREFETCH]
        for op in ops:
            for t in op.keep:
                if t is not None:
                    self._recorded.setdefault(t.data_ptr(), t)
        i = 0
        while i < len(ops):
            op = ops[i]
            nxt = ops[i + 1] if i + 1 < len(ops) else None
            if op.type == TaskType.RMS_NORM and nxt is not None and nxt.type == TaskType.LINEAR and self._norm_linear(op, nxt):
                i += 2
            elif op.type == TaskType.QK_NORM_ROPE_KV:
                if not (i + 2 < len(ops) and ops[i + 1].type == TaskType.ATTN_SPLIT and ops[i + 2].type == TaskType.ATTN_COMBINE):
                    raise LoweringError(f"{op.name}: a rope / cache task must be followed by its attention")
                self._attention(op, ops[i + 1], ops[i + 2])
                i += 3
            elif op.type == TaskType.LINEAR:
                self._linear(op)
                i += 1
            elif op.type == TaskType.RMS_NORM:
                self._rms_norm(op)
                i += 1
            elif op.type == TaskType.ADD:
                a, b_, out = op.keep[:3]
                self._emit(op.name, lambda vals, a=a, b_=b_, out=out: self._set(vals, out, torch.add(self._get(vals, a), self._get(vals, b_))))
                i += 1
            elif op.type == TaskType.SILU_MUL:
                self._silu_mul(op)
                i += 1
            elif op.type == TaskType.ALLREDUCE:
                self._allreduce(op)
                i += 1
            else:
                raise LoweringError(f"{op.name}: task type {op.type.name} has no graph lowering")
        written = {w for op in ops for w in op.writes if w in self._recorded}
        self._outputs = sorted(written if self._wanted is None else written & self._wanted)

    def _set(self, vals, recorded: torch.Tensor, value: torch.Tensor) -> None:
        vals[recorded.data_ptr()] = value

    def _norm_linear(self, norm: OpRecord, lin: OpRecord) -> bool:
        from ..ops.gemm import gemm_nt_norm, norm_gemm_supported

        x, nw, nout, res, res_out = norm.keep[:5]
        lx, w, lout, lres = lin.keep[:4]
        mode = lin.ints[7]
        if lx.data_ptr() != nout.data_ptr() or lres is not None:
            return False
        if self._readers_until_rewrite(nout.data_ptr(), norm.op_id) != [lin.op_id]:
            return False
        if not (x.is_cuda and norm_gemm_supported(x, w, nw, res)):
            return False
        eps = norm.floats[0]

        def step(vals):
            out, hidden = gemm_nt_norm(self._get(vals, x), nw, eps, w, residual=self._get(vals, res), silu=mode == 1)
            self._set(vals, lout, out)
            if res_out is not None:
                self._set(vals, res_out, hidden)

        self._emit(f"{norm.name}+{lin.name}", step)
        return True

    def _linear(self, op: OpRecord) -> None:
        from ..ops.gemm import gemm_nt, gemm_nt_silu_mul

        x, w, out, res = op.keep[:4]
        mode = op.ints[7]

        def step(vals):
            a = self._get(vals, x)
            if mode == 1:
                self._set(vals, out, gemm_nt_silu_mul(a, w))
                return
            y = gemm_nt(a, w)
            if res is not None:
                y = y + self._get(vals, res)
            self._set(vals, out, y)

        self._emit(op.name, step)

    def _rms_norm(self, op: OpRecord) -> None:
        from ..ops.elementwise import rms_norm

        x, nw, out, res, res_out = op.keep[:5]
        eps = op.floats[0]

        def step(vals):
            if res is not None:
                y, hidden = rms_norm(self._get(vals, x), nw, eps, residual=self._get(vals, res))
                if res_out is not None:
                    self._set(vals, res_out, hidden)
            else:
                y = rms_norm(self._get(vals, x), nw, eps)
            self._set(vals, out, y)

        self._emit(op.name, step)

    def _silu_mul(self, op: OpRecord) -> None:
        from ..ops.elementwise import silu_mul

        gate_up, out = op.keep[:2]
        self._emit(op.name, lambda vals: self._set(vals, out, silu_mul(self._get(vals, gate_up))))

    def _attention(self, rope: OpRecord, split: OpRecord, comb: OpRecord) -> None:
        from ..ops.flash_decode import decode_step_from_qkv, decode_step_from_qkv_supported

        qkv, q_w, k_w, cos, sin, kv_lens, kc, vc, table, q_out = rope.keep[:10]
        query, kc2, vc2, table2, kv_lens2, _partial, out = split.keep[:7]
        B, Hq, Hkv, D, _splits, page, _stride, len_is_pos = split.ints[:8]
        scale, soft_cap = split.floats[:2]
        if query.data_ptr() != q_out.data_ptr() or kc2.data_ptr() != kc.data_ptr() or kv_lens2.data_ptr() != kv_lens.data_ptr():
            raise LoweringError(f"{rope.name}: the attention that follows reads other tensors")
        pages = kc.shape[0]
        ident = pages % B == 0 and table.shape == (B, pages // B) and bool(
            torch.equal(table.cpu(), torch.arange(pages, dtype=torch.int32).view(B, -1)))
        kcv = kc.view(B, (pages // B) * page, Hkv, D) if ident else None
        vcv = vc.view(B, (pages // B) * page, Hkv, D) if ident else None
        qkv3 = qkv.view(B, 1, -1)
        if not (ident and len_is_pos and soft_cap == 0.0 and decode_step_from_qkv_supported(qkv3, kcv, vcv, D, cos)):
            raise LoweringError(f"{rope.name}: needs head_dim 128, bf16, an identity page table and no soft cap")
        eps = rope.floats[0]
        if B not in self._total_scratch:
            self._total_scratch[B] = torch.zeros(B, dtype=torch.int32, device=qkv.device)
            self._seq_tables[B] = torch.arange(B, dtype=torch.int32, device=qkv.device).view(B, 1)
        total, seq_table = self._total_scratch[B], self._seq_tables[B]
        # the rotation position of the new token = its cache slot; one int64 copy per step serves every layer
        if kv_lens.data_ptr() not in self._pos64:
            self._pos64[kv_lens.data_ptr()] = (kv_lens, torch.zeros(kv_lens.numel(), dtype=torch.int64, device=qkv.device))
        pos = self._pos64[kv_lens.data_ptr()][1].view(B, 1)

        def step(vals):
            o = decode_step_from_qkv(self._get(vals, qkv).view(B, 1, -1), q_w, k_w, eps, eps, pos, cos, sin,
                                     kcv, vcv, kv_lens, total, Hq, Hkv, seq_table=seq_table, scale=scale)
            self._set(vals, out, o)

        self._emit(f"{rope.name}+{split.name}+{comb.name}", step)

    def _allreduce(self, op: OpRecord) -> None:
        from ..ops.allreduce import all_reduce, create_allreduce_ctx

        x, out, _flags = op.keep[:3]
        b = self.b
        if self._ar_ctx is None:
            nbytes = max(o.keep[0].numel() * o.keep[0].element_size() for o in b.ops if o.type == TaskType.ALLREDUCE)
            self._ar_ctx = create_allreduce_ctx(2 * nbytes * b.world_size + 4096, b.rank, b.world_size, env=b.env)
        ctx = self._ar_ctx

        def step(vals):
            src = self._get(vals, x)
            self._set(vals, out, all_reduce(src.contiguous(), ctx=ctx))

        self._emit(op.name, step)

    # ---- execution ----------------------------------------------------------------------------------------------
    def run_eager(self) -> None:
        vals: Dict[int, torch.Tensor] = {}
        for lens, pos in self._pos64.values():
            pos.copy_(lens)
        for step in self.steps:
            step(vals)
        for ptr in self._outputs:
            v = vals.get(ptr)
            rec = self._recorded[ptr]
            if v is not None and v.data_ptr() != ptr:
                rec.copy_(v.view(rec.shape))

    def capture(self) -> None:
        """Record the step into a graph on a stream of its own.  Unlike ``torch.cuda.graph`` this does not
        synchronise the device first: with several ranks on one GPU (virtual ranks) a peer's replayed step may be
        spinning on this rank's next launch at this very moment, and recording executes nothing anyway."""
        g = torch.cuda.CUDAGraph()
        if self._capture_stream is None:
            self._capture_stream = torch.cuda.Stream()
        with torch.cuda.stream(self._capture_stream):
            g.capture_begin()
            try:
                self.run_eager()
            finally:
                g.capture_end()
        self.graph = g

    def run(self) -> None:
        """Call 1 runs the launches eagerly (kernels that size workspaces on first use must not meet a capture
        first; at world size > 1 every rank is inside the same call, so the peers' kernels are there), call 2
        captures the step - nothing executes during a capture, so ranks may capture at different times - and replays
        it, later calls replay."""
        self.calls += 1
        if self.calls == 1 or not torch.cuda.is_available():
            self.run_eager()
            return
        if self.graph is None:
            self.capture()
        self.graph.replay()
