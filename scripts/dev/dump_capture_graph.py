"""Capture (never replay) one decode step of the tiny dense model in ``triton_dist`` mode on two virtual
ranks and write the captured HIP graphs as DOT files: node kinds, edges and launch parameters of what a
replay would execute.  Nothing captured is executed, so this cannot fault the device."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["TDIST_TRACE_LAUNCH"] = "1"
os.environ["TDIST_ALLOW_STRADDLE_CAPTURE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

from triton_distributed_gh_amd.models import AutoLLM  # noqa: E402
from triton_distributed_gh_amd.models.config import ModelConfig  # noqa: E402
from triton_distributed_gh_amd.models.kv_cache import KV_Cache  # noqa: E402
from triton_distributed_gh_amd.utils.testing import VirtualCluster  # noqa: E402

out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r2"
os.makedirs(out_dir, exist_ok=True)
BF = torch.bfloat16
name, w, bsz, prompt, L = "tdist/tiny-dense", 2, 4, 16, 128
per = bsz // w
sh = lambda t, r: t[r * per:(r + 1) * per].contiguous()   # noqa: E731
torch.manual_seed(7)
ids = torch.randint(0, 1000, (bsz, prompt), device="cuda")
pos = torch.arange(prompt, device="cuda")[None].repeat(bsz, 1)
tok = torch.randint(0, 1000, (bsz, 1), device="cuda")
vc = VirtualCluster(w, heap_bytes=1 << 28)
models = [AutoLLM.from_pretrained(ModelConfig(model_name=name, max_length=L, dtype=BF, rank=r, world_size=w),
                                  env=vc.envs[r]) for r in range(w)]
kvs = [KV_Cache(models[0].num_layers, bsz, L, models[0].num_key_value_heads, models[0].head_dim, BF, w) for _ in range(w)]
for m in models:
    m.set_fwd("triton_dist")
vc.collective(lambda r, env: models[r].init_triton_dist_ctx(max_M=bsz * prompt))
torch.cuda.synchronize()
vc.run(lambda r, env: models[r].inference(sh(ids, r), pos, kvs[r]))            # eager prefill
for kv in kvs:
    kv.kv_offset.fill_(prompt)
s_tok = [sh(tok, r).clone() for r in range(w)]
s_pos = [kvs[r].get_kv_len()[:, None].long().clone() for r in range(w)]
torch.cuda.synchronize()
outs = vc.run(lambda r, env: models[r].inference(s_tok[r], s_pos[r], kvs[r]))  # eager decode warm-up
print("[dump] eager decode ok, logits", [tuple(o.shape) for o in outs], flush=True)
for r in range(w):
    a, b = torch.cuda.mem_get_info()
    print(f"[dump] capturing rank {r}", flush=True)
    sys.stderr.write(f"[dump] ---- capture rank {r} ----\n")
    gr = torch.cuda.CUDAGraph()
    gr.enable_debug_mode()
    with torch.cuda.graph(gr, stream=vc.streams[r]):
        logits = models[r].inference(s_tok[r], s_pos[r], kvs[r])
    path = os.path.join(out_dir, f"decode_graph_rank{r}.dot")
    gr.debug_dump(path)
    print(f"[dump] wrote {path} ({os.path.getsize(path)} bytes); logits {logits.data_ptr():#x}", flush=True)
torch.cuda.synchronize()
print("[dump] done", flush=True)
os._exit(0)      # graphs in debug mode are not torn down; leave without running destructors
