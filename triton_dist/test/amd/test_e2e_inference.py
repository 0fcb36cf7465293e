"""Engine generation: torch prefill, then decode with the selected back-end.

    bash scripts/launch_amd.sh python/triton_dist/test/amd/test_e2e_inference.py --bsz 4096 --gen_len 128 --triton_dist
    python triton_dist/test/amd/test_e2e_inference.py --model tdist/tiny-dense --bsz 4 --gen_len 8

(CLI of reference python/triton_dist/test/amd/test_e2e_inference.py:37-47 plus the NVIDIA script's --backend.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
os.environ.setdefault("RANDOM_PARAMS", "1")

import torch  # noqa: E402

from triton_dist.models import ModelConfig  # noqa: E402
from triton_dist.models.engine import Engine  # noqa: E402
from triton_dist.utils import finalize_distributed  # noqa: E402
from triton_distributed_gh_amd.utils import initialize_distributed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bsz", type=int, default=4096)
    ap.add_argument("--gen_len", type=int, default=128)
    ap.add_argument("--max_length", type=int, default=150)
    ap.add_argument("--prompt_len", type=int, default=16)
    ap.add_argument("--model", default="Qwen/Qwen3-32B")
    ap.add_argument("--num_layers", type=int, default=None)
    ap.add_argument("--triton_dist", action="store_true")
    ap.add_argument("--backend", default=None, choices=["torch", "triton_dist", "triton_dist_AR", "triton_dist_gemm_ar"])
    ap.add_argument("--no_graph", action="store_true")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--temperature", type=float, default=0.0)
    args = ap.parse_args()
    env = initialize_distributed(seed=0)
    cfg = ModelConfig(model_name=args.model, max_length=max(args.max_length, args.prompt_len + args.gen_len + 2),
                      dtype=torch.bfloat16, rank=env.rank, world_size=env.world_size, num_layers=args.num_layers)
    eng = Engine(cfg, temperature=args.temperature, top_p=0.95, group=env.tp_group, env=env)
    eng.backend = args.backend or ("triton_dist" if args.triton_dist else "torch")
    eng.no_graph = args.no_graph
    eng.enable_profile = args.profile
    torch.manual_seed(0)
    prompt = torch.randint(0, 1000, (args.bsz, args.prompt_len))
    out = eng.serve(prompt, args.gen_len)
    if env.rank == 0:
        print(f"generated {tuple(out.shape)} tokens in {eng.last_decode_seconds:.3f} s "
              f"({eng.last_decode_seconds / args.gen_len * 1e3:.3f} ms/step, backend {eng.backend})", flush=True)
    finalize_distributed()


if __name__ == "__main__":
    main()
