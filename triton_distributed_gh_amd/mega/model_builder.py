"""ModelBuilder: record ops (``make_*``), derive tile tasks + dependencies, schedule them onto work
queues and run the whole graph as ONE persistent kernel launch (``compile(backend="interpreter")``), or lower the
same op list onto the library's fused launches inside ONE HIP graph (``compile(backend="graph")``,
mega/graph_backend.py - the faster of the two on MI355X).

Reference: python/triton_dist/mega_triton_kernel/models/model_builder.py:85-404 (``make_fc1 / fc2 /
qkv_proj / o_proj / linear`` :189-201, ``make_attn`` :211, ``make_silu_mul_up`` :251,
``make_qk_norm_rope_update_kvcache`` :259, ``make_rms_norm`` :299, ``make_add`` :314,
``make_barrier_all_intra_node`` :326, ``make_allreduce`` :331, ``make_prefetch`` :352, ``compile``
:372-389, ``run`` :391-404).  Dependencies are derived from the tensors: an op depends on the last
writer of every tensor it reads (whole-op tile ranges, or the matching row / token tile where the
producer is tiled the same way).
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import torch

from .. import _lib
from ..utils.distributed import get_env, is_initialized
from .core import (MAX_DEPS, SCHEDULERS, MegaOpStruct, OpRecord, Task, TaskDependency, TaskType, check_schedule,
                   encode_queues, register_task)


def _linear_bn(N: int, num_queues: int) -> int:
    """Columns per linear task.  Fewer, larger tasks amortise the per-task overhead; more tasks keep
    every CU streaming: take the largest BN that still yields >= 256 tasks (one per CU)."""
    for bn in (64, 32, 16):
        if (N + bn - 1) // bn >= 256 or bn == 16:
            return bn
    return 16


class ModelBuilder:

    def __init__(self, rank: int = 0, world_size: int = 1, local_world_size: int = 1, num_queues: Optional[int] = None,
                 scheduler: str = "round_robin", env=None):
        self.rank, self.world_size, self.local_world_size = rank, world_size, local_world_size
        self.env = env or (get_env() if is_initialized() else None)
        self.device = self.env.device if self.env is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self.num_queues = num_queues or 256
        self.scheduler = scheduler
        self.ops: List[OpRecord] = []
        self._last_writer: Dict[int, int] = {}      # tensor data_ptr -> op id
        self._compiled = None
        self._epoch = 0
        self._metrics = {"memory": 0}
        self.logger_enabled = False

    # ---- bookkeeping ---------------------------------------------------------------------------
    def create_symm_tensor(self, shape, dtype) -> torch.Tensor:
        assert self.env is not None, "symmetric tensors need initialize_distributed() / a virtual env"
        t = self.env.heap.create_tensor(list(shape), dtype)
        self._metrics["memory"] += t.numel() * t.element_size()
        return t

    def _add(self, type_, name, num_tiles, ptrs, ints, floats, reads, writes, keep) -> OpRecord:
        op = OpRecord(len(self.ops), type_, name, int(num_tiles), [int(p) for p in ptrs], [int(i) for i in ints],
                      [float(f) for f in floats], list(keep), [w.data_ptr() for w in writes],
                      [r.data_ptr() for r in reads if r is not None])
        self.ops.append(op)
        self._compiled = None
        return op

    def _commit(self, op: OpRecord) -> None:
        for w in op.writes:
            self._last_writer[w] = op.op_id

    @staticmethod
    def _ptr(t: Optional[torch.Tensor]) -> int:
        return 0 if t is None else t.data_ptr()

    # ---- op constructors --------------------------------------------------------------------------
    def make_linear(self, x, weight, out, residual=None, mode: int = 0, name="linear", config=None):
        """out[M, N] (+ residual) = x[M, K] @ weight[N, K]^T.  ``mode=1``: weight is [2N, K]
        (gate rows then up rows) and out = silu(gate) * up.  ``config``: a ``tasks.LinearConfig`` whose
        ``BLOCK_SIZE_N`` (16 / 32 / 64) fixes the columns per task."""
        M, K = x.shape
        N = weight.shape[0] // (2 if mode == 1 else 1)
        assert M <= 16, "mega-kernel linear tasks are decode-shaped (at most 16 tokens)"
        assert K % 32 == 0 and weight.shape[1] == K and out.shape == (M, N)
        assert x.dtype == weight.dtype == out.dtype == torch.bfloat16
        assert x.stride(1) == 1 and weight.is_contiguous() and out.stride(1) == 1
        from .tasks import linear_block_n

        bn = linear_block_n(config) or _linear_bn(N, self.num_queues)
        op = self._add(TaskType.LINEAR, name, (N + bn - 1) // bn, [x.data_ptr(), weight.data_ptr(), out.data_ptr(), self._ptr(residual)],
                       [M, N, K, x.stride(0), out.stride(0), residual.stride(0) if residual is not None else 0, bn, mode],
                       [], [x, residual], [out], [x, weight, out, residual])
        self._commit(op)
        return op

    def make_fc1(self, x, gate_up_weight, out, name="mlp_fc1", config=None):
        return self.make_linear(x, gate_up_weight, out, mode=1, name=name, config=config)

    def make_fc2(self, x, weight, out, residual=None, name="mlp_fc2", config=None):
        return self.make_linear(x, weight, out, residual, name=name, config=config)

    def make_qkv_proj(self, x, weight, out, name="qkv_proj", config=None):
        return self.make_linear(x, weight, out, name=name, config=config)

    def make_o_proj(self, x, weight, out, residual=None, name="o_proj", config=None):
        return self.make_linear(x, weight, out, residual, name=name, config=config)

    def make_rms_norm(self, x, weight, out, eps=1e-6, residual=None, residual_out=None, name="rms_norm"):
        M, H = x.shape
        assert H % 8 == 0 and x.is_contiguous() and out.is_contiguous()
        op = self._add(TaskType.RMS_NORM, name, M,
                       [x.data_ptr(), weight.data_ptr(), out.data_ptr(), self._ptr(residual), self._ptr(residual_out)],
                       [M, H], [eps], [x, residual], [out] + ([residual_out] if residual_out is not None else []),
                       [x, weight, out, residual, residual_out])
        self._commit(op)
        return op

    def make_add(self, a, b, out, name="add"):
        M, H = a.shape
        assert H % 8 == 0 and a.is_contiguous() and b.is_contiguous() and out.is_contiguous()
        op = self._add(TaskType.ADD, name, M, [a.data_ptr(), b.data_ptr(), out.data_ptr()], [M, H], [], [a, b], [out],
                       [a, b, out])
        self._commit(op)
        return op

    def make_silu_mul_up(self, gate_up, out, name="silu_mul_up"):
        M, I2 = gate_up.shape
        assert I2 % 16 == 0 and gate_up.is_contiguous() and out.is_contiguous()
        op = self._add(TaskType.SILU_MUL, name, M, [gate_up.data_ptr(), 0, out.data_ptr()], [M, I2 // 2], [], [gate_up],
                       [out], [gate_up, out])
        self._commit(op)
        return op

    def make_qk_norm_rope_update_kvcache(self, qkv, q_norm_w, k_norm_w, cos, sin, kv_lens, key_cache, value_cache,
                                         block_tables, q_out, eps=1e-6, name="qk_norm_rope_update_kvcache"):
        """qkv [M, (Hq + 2 Hkv) * D] -> q_out [M, Hq, D] (normed + rotated) and the new token's K / V
        written into the paged cache at position kv_lens[b]."""
        M = qkv.shape[0]
        _, page, Hkv, D = key_cache.shape
        Hq = q_out.shape[1]
        assert qkv.shape[1] == (Hq + 2 * Hkv) * D and D in (64, 128, 256) and cos.dtype == torch.float32
        assert kv_lens.dtype == torch.int32 and block_tables.dtype == torch.int32
        op = self._add(TaskType.QK_NORM_ROPE_KV, name, M,
                       [qkv.data_ptr(), self._ptr(q_norm_w), self._ptr(k_norm_w), cos.data_ptr(), sin.data_ptr(),
                        kv_lens.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), block_tables.data_ptr(),
                        q_out.data_ptr()],
                       [M, Hq, Hkv, D, page, block_tables.stride(0)], [eps], [qkv], [q_out, key_cache, value_cache],
                       [qkv, q_norm_w, k_norm_w, cos, sin, kv_lens, key_cache, value_cache, block_tables, q_out])
        self._commit(op)
        return op

    def make_attn(self, query, key_cache, value_cache, block_tables, kv_lens, out, partial=None, num_splits: int = 0,
                  sm_scale: Optional[float] = None, soft_cap: float = 0.0, kv_len_is_position: bool = True,
                  name="attn"):
        """Decode attention = split-KV tasks + combine tasks.  query [B, Hq, D], out [B, Hq, D]."""
        B, Hq, D = query.shape
        _, page, Hkv, _ = key_cache.shape
        assert D in (64, 128), "the mega-kernel attention task supports head_dim 64 / 128"
        group = Hq // Hkv
        Y = Hkv * ((group + 15) // 16)
        max_len = block_tables.shape[1] * page
        if num_splits <= 0:
            num_splits = max(1, min(16, max_len // 512, max(1, self.num_queues // max(1, B * Y))))
        if partial is None:
            partial = torch.empty(B, Hq, num_splits, D + 1, dtype=torch.float32, device=query.device)
        scale = sm_scale if sm_scale is not None else D ** -0.5
        ptrs = [query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), block_tables.data_ptr(), kv_lens.data_ptr(),
                partial.data_ptr(), out.data_ptr()]
        ints = [B, Hq, Hkv, D, num_splits, page, block_tables.stride(0), 1 if kv_len_is_position else 0]
        keep = [query, key_cache, value_cache, block_tables, kv_lens, partial, out]
        split = self._add(TaskType.ATTN_SPLIT, name + "_split", B * Y * num_splits, ptrs, ints, [scale, soft_cap],
                          [query, key_cache, value_cache], [partial], keep)
        self._commit(split)
        comb = self._add(TaskType.ATTN_COMBINE, name + "_combine", B, ptrs, ints, [scale, soft_cap], [partial], [out], keep)
        self._commit(comb)
        return split, comb

    def make_allreduce(self, input, output, double_input_buffer: bool = False, name="allreduce"):
        """``input`` must be a symmetric tensor [M, H]; output is local.  One-shot pull reduce."""
        assert self.env is not None and self.world_size > 1
        M, H = input.shape
        flags = self.env.heap.create_tensor([M * self.world_size], torch.int32, arena="sig")
        op = self._add(TaskType.ALLREDUCE, name, M, [input.data_ptr(), output.data_ptr(), flags.data_ptr()], [M, H], [],
                       [input], [output], [input, output, flags])
        self._commit(op)
        return op

    def make_barrier_all_intra_node(self, name="barrier_all_intra_node"):
        assert self.env is not None
        slots = self.env.heap.create_tensor([max(self.world_size, 2)], torch.int32, arena="sig")
        op = self._add(TaskType.BARRIER, name, 1, [slots.data_ptr()], [], [], [], [], [slots])
        # a barrier orders everything recorded before it: depend on every earlier op
        op.reads = [w for prev in self.ops[:-1] for w in prev.writes][-MAX_DEPS:]
        return op

    def make_prefetch(self, weight, chunk_bytes: int = 1 << 20, name="prefetch"):
        total = weight.numel() * weight.element_size()
        op = self._add(TaskType.PREFETCH, name, (total + chunk_bytes - 1) // chunk_bytes,
                       [weight.data_ptr()], [chunk_bytes, total & 0xFFFFFFFF, total >> 32], [], [], [], [weight])
        return op

    def reset(self):
        self.ops.clear()
        self._last_writer.clear()
        self._compiled = None

    # ---- tasks + dependencies ------------------------------------------------------------------------
    def _writer_of(self, ptr: int, before: int) -> Optional[OpRecord]:
        for op in reversed(self.ops[:before]):
            if ptr in op.writes:
                return op
        return None

    def build_tasks(self) -> List[Task]:
        """Tile tasks with their dependencies.  An op waits for the last writer of everything it reads and - because
        recorded tensors are reused (one normalised-activation buffer serves every norm of a model) - for the earlier
        readers and the earlier writer of everything it WRITES, unless a dependency it already has orders it behind
        them: ``settled[w]`` is the set of ops that are complete once every tile of ``w`` is, and a full-range
        dependency on ``w`` puts all of them in front of every tile of the waiting op."""
        tasks: List[Task] = []
        row_tiled = (TaskType.RMS_NORM, TaskType.ADD, TaskType.SILU_MUL, TaskType.QK_NORM_ROPE_KV, TaskType.ALLREDUCE,
                     TaskType.ATTN_COMBINE)

        def tile_dep(op, w, tile):
            if op.type in row_tiled and w.type in row_tiled and w.num_tiles == op.num_tiles:
                return TaskDependency(w.op_id, tile, tile + 1)          # same row / token
            if op.type == TaskType.ATTN_SPLIT and w.type == TaskType.QK_NORM_ROPE_KV:
                b = tile // (op.num_tiles // w.num_tiles)
                return TaskDependency(w.op_id, b, b + 1)
            if op.type == TaskType.ATTN_COMBINE and w.type == TaskType.ATTN_SPLIT:
                per = w.num_tiles // op.num_tiles
                return TaskDependency(w.op_id, tile * per, (tile + 1) * per)
            return TaskDependency(w.op_id, 0, w.num_tiles)

        def full_range(op, w):
            d = tile_dep(op, w, 0)
            return d.lo == 0 and d.hi == w.num_tiles

        settled: List[set] = []                 # per op: ops complete once all of its tiles are
        readers: Dict[int, List[int]] = {}      # tensor -> ops that read it since its last write
        writer: Dict[int, int] = {}             # tensor -> last op that wrote it
        for op in self.ops:
            producers = []
            for ptr in dict.fromkeys(op.reads):
                w = self._writer_of(ptr, op.op_id)
                if w is not None and w.op_id not in [p.op_id for p in producers]:
                    producers.append(w)
            # everything in front of EVERY tile of this op through the dependencies it has so far
            ahead = set()
            for w in producers:
                if full_range(op, w):
                    ahead |= {w.op_id} | settled[w.op_id]
            hazards = []
            for ptr in dict.fromkeys(op.writes):
                hazards += readers.get(ptr, [])
                if ptr in writer:
                    hazards.append(writer[ptr])
            for hz in sorted(set(hazards), reverse=True):          # latest first: it may settle the earlier ones
                if hz == op.op_id or hz in ahead or hz in [p.op_id for p in producers]:
                    continue
                w = self.ops[hz]
                producers.append(w)
                if full_range(op, w):
                    ahead |= {hz} | settled[hz]
            assert len(producers) <= MAX_DEPS, f"{op.name}: more than {MAX_DEPS} dependencies"
            done = set()
            for w in producers:
                done |= {w.op_id} | settled[w.op_id]
            settled.append(done)
            for ptr in dict.fromkeys(op.reads):
                readers.setdefault(ptr, []).append(op.op_id)
            for ptr in dict.fromkeys(op.writes):
                readers[ptr] = []
                writer[ptr] = op.op_id
            for tile in range(op.num_tiles):
                tasks.append(Task(op.type, op.op_id, tile, [tile_dep(op, w, tile) for w in producers]))
        return tasks

    # ---- compile / run -------------------------------------------------------------------------------
    def compile(self, validate: bool = True, backend: str = "interpreter", outputs=None):
        """``backend``: "interpreter" (one persistent kernel walking the task queues), "graph" (fused native
        launches captured in one HIP graph; raises ``LoweringError`` for a graph it cannot lower) or "auto" (graph
        when the op list lowers, else the interpreter).  ``outputs`` (graph back-end): the recorded tensors the caller
        reads after a step - only those are copied back from the fused launches' own buffers; default: every
        tensor the graph writes."""
        assert backend in ("interpreter", "graph", "auto")
        if backend != "interpreter" and self.device.type == "cuda":
            from .graph_backend import GraphProgram, LoweringError

            try:
                prog = GraphProgram(self, outputs)
            except LoweringError:
                if backend == "graph":
                    raise
                prog = None
            if prog is not None:
                self._compiled = dict(backend="graph", program=prog, num_tasks=len(prog.steps), err=None)
                return self._compiled
        elif backend == "graph":
            raise RuntimeError("the graph back-end needs a GPU")
        tasks = self.build_tasks()
        queues = SCHEDULERS[self.scheduler](tasks, self.num_queues)
        if validate:
            check_schedule(queues)
        qbuf, qlen, max_tasks = encode_queues(queues, self.device)
        arr = (MegaOpStruct * len(self.ops))(*[op.pack() for op in self.ops])
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        max_tiles = max(op.num_tiles for op in self.ops)
        scoreboard = torch.zeros(len(self.ops) * max_tiles, dtype=torch.int32, device=self.device)
        err = torch.zeros(4, dtype=torch.int32, device=self.device)
        self._compiled = dict(backend="interpreter", queues=qbuf, qlen=qlen, max_tasks=max_tasks, ops=raw,
                              scoreboard=scoreboard, err=err, max_tiles=max_tiles, num_tasks=len(tasks))
        self._metrics["memory"] += qbuf.numel() * 4 + raw.numel() + scoreboard.numel() * 4
        return self._compiled

    def run(self, stream=None):
        c = self._compiled or self.compile()
        if c["backend"] == "graph":
            c["program"].run()
            return
        self._epoch += 1
        cs = self.env.heap.ctx_struct() if self.env is not None else _null_ctx()
        _lib.check(
            _lib.hip().tdist_mega_launch(ctypes.byref(cs), _lib.ptr(c["queues"]), _lib.ptr(c["qlen"]), _lib.ptr(c["ops"]),
                                         _lib.ptr(c["scoreboard"]), _lib.ptr(c["err"]), self.num_queues, c["max_tasks"],
                                         c["max_tiles"], self._epoch, _lib.stream_handle(stream)), "tdist_mega_launch")

    def check_error(self):
        if self._compiled is not None and self._compiled["backend"] == "graph":
            ctx = self._compiled["program"]._ar_ctx
            if ctx is not None:
                ctx.check_error()
            return
        if self._compiled is not None and int(self._compiled["err"][0]) != 0:
            raise RuntimeError(f"mega kernel reported error word {int(self._compiled['err'][0])} "
                               "(1 = dependency wait timed out, 2 = barrier timed out, 4 = unknown task type)")

    def memory_bytes(self) -> int:
        return self._metrics["memory"]


def _null_ctx():
    from ..utils.symm import SymmCtxStruct

    c = SymmCtxStruct()
    c.rank, c.num_ranks = 0, 1
    return c


# task-type registry entries (the reference registers 14 op types, tasks/*.py)
for _name, _fn in [("linear", ModelBuilder.make_linear), ("mlp_fc1", ModelBuilder.make_fc1), ("mlp_fc2", ModelBuilder.make_fc2),
                   ("qkv_proj", ModelBuilder.make_qkv_proj), ("o_proj", ModelBuilder.make_o_proj),
                   ("silu_mul_up", ModelBuilder.make_silu_mul_up),
                   ("qk_norm_rope_update_kvcache", ModelBuilder.make_qk_norm_rope_update_kvcache),
                   ("rms_norm", ModelBuilder.make_rms_norm), ("attn_split", ModelBuilder.make_attn),
                   ("attn_combine", ModelBuilder.make_attn), ("add", ModelBuilder.make_add),
                   ("allreduce", ModelBuilder.make_allreduce),
                   ("barrier_all_intra_node", ModelBuilder.make_barrier_all_intra_node),
                   ("prefetch", ModelBuilder.make_prefetch)]:
    register_task(_name)(_fn)
