"""Inference engine: torch prefill, then a decode loop replayed from one captured HIP graph.

Reference: python/triton_dist/models/engine.py:36-189.  Back-ends: ``torch`` (RCCL all-reduce),
``triton_dist`` (AG+GEMM / GEMM+RS, batch sharded over ranks), ``triton_dist_AR`` (symmetric-memory
all-reduce kernels) and ``triton_dist_gemm_ar`` (GEMM fused with the all-reduce).

Graph replay is used where the whole step is replay-safe: the torch back-end, any back-end at
world size 1, and at any world size ``triton_dist_AR``, ``triton_dist`` (dense and MoE models) and
``triton_dist_gemm_ar``, at any decode batch: the all-reduce kernels, AG + GEMM, GEMM + RS (both tile forms), the
MoE pair (AG + grouped GEMM, grouped GEMM + top-k reduce + RS, with the low-latency all-gather of the routing data)
and both forms of GEMM + all-reduce keep their call number on the device, so a replayed launch picks a fresh epoch
and buffer parity by itself (csrc/include/tdist/call_state.hpp).
"""
from __future__ import annotations

import time
from typing import List, Optional

import torch
import torch.distributed as dist

from ..ops.allreduce import AllReduceMethod
from . import AutoLLM, AutoTokenizer
from .config import ModelConfig
from .kv_cache import KV_Cache
from .utils import logger, sample_token


class Engine:

    def __init__(self, model_config: ModelConfig, temperature: float = 0.0, top_p: float = 0.95, verbose: bool = False,
                 group=None, env=None):
        self.logger = logger
        self.model_config, self.group, self.env = model_config, group, env
        self.temperature, self.top_p, self.verbose = temperature, top_p, verbose
        self.kv_cache: Optional[KV_Cache] = None
        self.no_graph = False
        self.backend = "torch"
        self.enable_profile = False
        self.last_decode_seconds = None
        self._init_model()

    def _init_model(self):
        self.logger.log(f"Initializing model {self.model_config.model_name}...", "info")
        self.model = AutoLLM.from_pretrained(self.model_config, self.group, env=self.env)
        self.tokenizer = AutoTokenizer.from_pretrained(self.model_config)
        self.logger.log("Model initialized!", "success")

    def _init_kv_cache(self, bsz: int):
        m = self.model
        self.kv_cache = KV_Cache(num_layers=m.num_layers, kv_heads=m.num_key_value_heads, head_dim=m.head_dim,
                                 batch_size=bsz, dtype=m.dtype, max_length=m.max_length, world_size=m.world_size,
                                 device=m.device)

    def _barrier(self):
        if self.model.device.type == "cuda":
            torch.cuda.synchronize()
        if dist.is_initialized():
            env = self.env
            dist.barrier(group=env.cpu_group if env is not None and env.cpu_group is not None else None)

    def _is_moe(self) -> bool:
        arch = getattr(self.model, "config", None)
        return bool(arch is not None and getattr(arch, "num_experts", 0) > 0)

    def _local_bsz(self, bsz: int) -> int:
        return bsz // self.model.world_size if self.backend == "triton_dist" else bsz

    def _init_cuda_graph(self, bsz: int = 1):
        """Capture ONE decode step; ``run`` refreshes the static inputs and replays it."""
        dev = self.model.device
        static_ids = torch.full((self._local_bsz(bsz), 1), 1, dtype=torch.long, device=dev)
        static_pos = torch.zeros((bsz, 1), dtype=torch.long, device=dev)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):      # warm-up: allocator, lazy tables, both parities of every context
                self.model.inference(input_ids=static_ids, position_ids=static_pos, kv_cache=self.kv_cache)
            s.synchronize()
        torch.cuda.current_stream().wait_stream(s)
        self._barrier()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            logits = self.model.inference(input_ids=static_ids, position_ids=static_pos, kv_cache=self.kv_cache)

        def run(input_ids, position_ids):
            static_ids.copy_(input_ids)
            static_pos.copy_(position_ids)
            graph.replay()
            return logits

        self._graph = graph
        self.logger.log("HIP graph captured!", "success")
        return run

    def get_ctx(self, input_ids: torch.Tensor, bsz: Optional[int] = None):
        past = self.kv_cache.get_kv_len()
        return past[:, None].long() + torch.arange(input_ids.size(1), device=past.device).long()

    def _switch_backend(self, bsz: int):
        m = self.model
        if self.backend == "triton_dist":
            assert bsz % m.world_size == 0, "triton_dist shards the batch over the ranks"
            m.set_fwd("triton_dist")
            m.init_triton_dist_ctx(max_M=bsz)
        elif self.backend == "triton_dist_AR":
            m.set_fwd("triton_dist_AR")
            m.init_triton_dist_AR_ctx(max_M=bsz, ar_method=AllReduceMethod.Unknown)
        elif self.backend == "triton_dist_gemm_ar":
            m.set_fwd("triton_dist_gemm_ar")
            m.init_triton_dist_gemm_ar_ctx(max_M=bsz)
        elif self.backend != "torch":
            raise ValueError(f"unknown backend {self.backend}")

    @torch.inference_mode()
    def serve(self, input_ids: torch.Tensor, gen_len: int) -> torch.Tensor:
        """Prefill ``input_ids [bsz, prompt]`` with the torch forward, then generate ``gen_len``
        tokens with the selected back-end.  Returns the generated ids [bsz (local), gen_len]."""
        m = self.model
        bsz = input_ids.shape[0]
        self.logger.log(f"Serving {m.model_name}: prefill {tuple(input_ids.shape)}, gen_len={gen_len}, "
                        f"backend={self.backend}", "info")
        self.kv_cache = None
        self._init_kv_cache(bsz)
        m.set_fwd("torch")
        ids = input_ids.to(m.device)
        logits = m.inference(input_ids=ids, position_ids=self.get_ctx(ids), kv_cache=self.kv_cache)
        next_token = sample_token(logits[:, -1, :], temperature=self.temperature, top_p=self.top_p)
        self.kv_cache.kv_offset.fill_(ids.shape[-1])

        self._switch_backend(bsz)
        if self.backend == "triton_dist":
            next_token = next_token.split(bsz // m.world_size, dim=0)[m.rank].contiguous()
        use_graph = not self.no_graph and m.device.type == "cuda" and (
            self.backend in ("torch", "triton_dist_AR", "triton_dist_gemm_ar", "triton_dist") or m.world_size == 1)
        if use_graph:
            self.model_launch = self._init_cuda_graph(bsz)
        else:
            self.model_launch = lambda i, p: m.inference(input_ids=i, position_ids=p, kv_cache=self.kv_cache)

        output_ids: List[torch.Tensor] = []
        profiler = None
        self._barrier()
        t0 = time.perf_counter()
        if self.enable_profile:
            profiler = torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU,
                                                          torch.profiler.ProfilerActivity.CUDA], record_shapes=True)
            profiler.__enter__()
        for step in range(gen_len):
            position_ids = self.kv_cache.get_kv_len()[:, None].long()
            logits = self.model_launch(next_token, position_ids)
            next_token = sample_token(logits[:, -1, :], temperature=self.temperature, top_p=self.top_p)
            self.kv_cache.inc_offset()
            output_ids.append(next_token)
            if profiler is not None and step + 1 >= min(64, gen_len):
                profiler.__exit__(None, None, None)
                profiler.export_chrome_trace("trace_static.json")
                profiler = None
        self._barrier()
        self.last_decode_seconds = time.perf_counter() - t0
        self.logger.log(f"Decoding finished! Total latency: {self.last_decode_seconds:.3f} s "
                        f"({self.last_decode_seconds / max(gen_len, 1) * 1e3:.3f} ms/step)")
        out = torch.cat(output_ids, dim=1).cpu()
        if self.verbose:
            print(self.tokenizer.batch_decode(out, skip_special_tokens=True))
        self.model_launch = None
        self._graph = None
        return out
