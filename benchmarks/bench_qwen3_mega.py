"""Single-step decode latency: torch eager / torch + HIP graph / native layers + HIP graph / mega kernel.

Scenario of the reference's mega_triton_kernel/test/models/bench_qwen3.py:113-153 (bs=1, ctx=512;
published numbers in docs/mega_triton_kernel.md:32-39 are for TP=8 on H800 / H20).  Random weights.

    python benchmarks/bench_qwen3_mega.py --model Qwen/Qwen3-8B --bsz 1 --ctx 512 [--layers N]
    torchrun --nproc-per-node 8 ... benchmarks/bench_qwen3_mega.py --model Qwen/Qwen3-32B
"""
import argparse
import json
import os
import sys

os.environ.setdefault("RANDOM_PARAMS", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from triton_distributed_gh_amd.mega import Qwen3Model  # noqa: E402
from triton_distributed_gh_amd.models import AutoLLM, KV_Cache, ModelConfig  # noqa: E402
from triton_distributed_gh_amd.utils import finalize_distributed, initialize_distributed  # noqa: E402
from triton_distributed_gh_amd.utils.perf import perf_func  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="Qwen/Qwen3-8B")
    ap.add_argument("--bsz", type=int, default=1)
    ap.add_argument("--ctx", type=int, default=512)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--queues", type=int, default=512)
    ap.add_argument("--scheduler", default="round_robin")
    args = ap.parse_args()
    env = initialize_distributed(seed=0, heap_bytes=1 << 28)
    L = args.ctx + 64
    cfg = ModelConfig(model_name=args.model, max_length=L, dtype=torch.bfloat16, rank=env.rank, world_size=env.world_size,
                      num_layers=args.layers)
    model = AutoLLM.from_pretrained(cfg, env.tp_group)
    B = args.bsz
    kv = KV_Cache(model.num_layers, B, L, model.num_key_value_heads, model.head_dim, torch.bfloat16, env.world_size)
    kv.rand_fill_kv_cache(args.ctx)
    kv.kv_offset.fill_(args.ctx)
    tok = torch.randint(0, 1000, (B, 1), device=env.device)
    pos = torch.full((B, 1), args.ctx, device=env.device, dtype=torch.long)
    model.set_fwd("torch")

    def eager():
        return model.inference(tok, pos, kv)

    _, t_eager = perf_func(eager, args.iters, 5)
    res = {"model": args.model, "layers": model.num_layers, "bsz": B, "ctx": args.ctx, "tp": env.world_size,
           "torch_eager_ms": round(t_eager, 4)}
    if env.world_size == 1:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                eager()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            eager()
        _, t_graph = perf_func(g.replay, args.iters, 5)
        res["torch_graph_ms"] = round(t_graph, 4)
        # same layer stack with every projection on this package's GEMM kernels (decode kernel for
        # M <= 16), one HIP graph per step: the reference's "triton_dist_AR + cudagraph" column
        model.set_fwd("triton_dist_AR")
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                eager()
        torch.cuda.current_stream().wait_stream(s)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            eager()
        _, t_native = perf_func(g2.replay, args.iters, 5)
        res["native_layers_graph_ms"] = round(t_native, 4)
        model.set_fwd("torch")
    # the mega API twice: its default back-end (the task graph lowered onto fused launches inside one HIP graph when
    # the model lowers, mega/graph_backend.py) and the persistent interpreter kernel
    t_mega = None
    # (the mega task bodies are decode-shaped, at most 16 tokens: larger batches time the layer paths only)
    for backend in (("auto", "interpreter") if B <= 16 else ()):
        mega = Qwen3Model(model, B, L, page_size=64, num_queues=args.queues, scheduler=args.scheduler, env=env,
                          backend=backend)
        mega.kv_cache.rand_fill_kv_cache(args.ctx)

        def mega_step():
            mega.set_kv_len(args.ctx)
            return mega.mega_forward(tok)

        _, t = perf_func(mega_step, args.iters, 5)
        mega.check_error()
        key = "mega_ms" if backend == "auto" else "mega_interpreter_ms"
        res[key] = round(t, 4)
        if backend == "auto":
            t_mega = t
            res["mega_backend"] = mega.backend
            res["mega_launches"] = mega.num_tasks
        else:
            res["mega_tasks"] = mega.num_tasks
        del mega
    wbytes = sum(t.numel() * t.element_size() for layer in model.layers
                 for t in (layer.attn.wqkv, layer.attn.wo, layer.mlp.gate_up_proj, layer.mlp.down_proj)) \
        + model.lm_head.numel() * 2
    res["weight_GB"] = round(wbytes / 1e9, 3)
    if t_mega is not None:
        res["mega_weight_stream_TBps"] = round(wbytes / (t_mega * 1e-3) / 1e12, 3)
    if env.rank == 0:
        print(json.dumps(res), flush=True)
    finalize_distributed()


if __name__ == "__main__":
    main()
