"""Prefill latency of one dense model on one rank (or one TP group): torch-mode layers (library GEMMs) and
this package's layers, both with the in-tree attention kernel.  Random weights, synthetic tokens.

Scenario of the reference's prefill column in docs/getting-started/e2e/e2e_dense.md:30-45 (bsz x seq_len
prompt through `model.inference`).

    python benchmarks/bench_prefill.py --model Qwen/Qwen3-8B --bsz 1 --seq 4096 [--layers N]
"""
import argparse
import json
import os
import sys

os.environ.setdefault("RANDOM_PARAMS", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from triton_distributed_gh_amd.models import AutoLLM, KV_Cache, ModelConfig  # noqa: E402
from triton_distributed_gh_amd.utils import finalize_distributed, initialize_distributed  # noqa: E402
from triton_distributed_gh_amd.utils.perf import perf_func  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="Qwen/Qwen3-8B")
    ap.add_argument("--bsz", type=int, default=1)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    env = initialize_distributed(seed=0, heap_bytes=1 << 30)
    cfg = ModelConfig(model_name=args.model, max_length=args.seq, dtype=torch.bfloat16, rank=env.rank,
                      world_size=env.world_size, num_layers=args.layers)
    model = AutoLLM.from_pretrained(cfg, env.tp_group)
    B, S = args.bsz, args.seq
    kv = KV_Cache(model.num_layers, B, S, model.num_key_value_heads, model.head_dim, torch.bfloat16, env.world_size)
    tok = torch.randint(0, 1000, (B, S), device=env.device)
    pos = torch.arange(S, device=env.device, dtype=torch.long).unsqueeze(0).expand(B, S).contiguous()
    res = {"model": args.model, "layers": model.num_layers, "bsz": B, "seq": S, "tp": env.world_size}

    def step():
        kv.clear()
        return model.inference(tok, pos, kv)

    modes = ["torch", "triton_dist_AR"] if env.world_size == 1 else ["torch", "triton_dist_AR", "triton_dist"]
    if hasattr(model.layers[0].mlp, "num_experts"):          # MoE layers have the triton_dist form only
        modes = ["torch", "triton_dist"]
    for mode in modes:
        if mode == "triton_dist_AR":
            model.init_triton_dist_AR_ctx(max_M=B * S)
        elif mode == "triton_dist":
            model.init_triton_dist_ctx(max_M=B * S)
        model.set_fwd(mode)
        _, ms = perf_func(step, args.iters, 3)
        res[f"{mode}_ms"] = round(ms, 3)
        res[f"{mode}_tokens_per_s"] = round(B * S / ms * 1e3)
    if env.rank == 0:
        print(json.dumps(res), flush=True)
    finalize_distributed()


if __name__ == "__main__":
    main()
