"""A few eager decode steps of a Qwen3-30B-A3B-shaped model (4 layers) for a kernel trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("RANDOM_PARAMS", "1")
from triton_distributed_gh_amd.models import AutoLLM, KV_Cache, ModelConfig
from triton_distributed_gh_amd.utils import finalize_distributed, initialize_distributed

env = initialize_distributed(seed=0, heap_bytes=1 << 28)
B, ctx, L = int(os.environ.get("BSZ", "1")), 128, 256
cfg = ModelConfig(model_name="Qwen/Qwen3-30B-A3B", max_length=L, dtype=torch.bfloat16, rank=0, world_size=1, num_layers=4)
model = AutoLLM.from_pretrained(cfg, env.tp_group)
kv = KV_Cache(model.num_layers, B, L, model.num_key_value_heads, model.head_dim, torch.bfloat16, 1)
kv.rand_fill_kv_cache(ctx)
kv.kv_offset.fill_(ctx)
model.set_fwd("triton_dist")
model.init_triton_dist_ctx(max_M=max(B, 16))
tok = torch.randint(0, 1000, (B, 1), device="cuda")
pos = torch.full((B, 1), ctx, device="cuda", dtype=torch.long)
for _ in range(12):
    model.inference(tok, pos, kv)
torch.cuda.synchronize()
finalize_distributed()
print("done")
