"""One Qwen3-30B-A3B MoE layer (128 experts, top-8, H 2048, I 768; 1.2 GB of expert weights) on one rank:
TP_MoE.dist_triton_fwd against the layer's own torch paths, 1 .. 512 tokens, us per call from HIP-graph replays."""
import os, sys, torch
from types import SimpleNamespace
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from triton_distributed_gh_amd.layers.tp_moe import TP_MoE
from triton_distributed_gh_amd.utils import initialize_distributed
from triton_distributed_gh_amd.utils.perf import graph_bench

BF = torch.bfloat16
E, topk, H, I = [int(v) for v in os.environ.get("SHAPE", "128,8,2048,768").split(",")]     # SHAPE=8,2,4096,14336: Mixtral-8x7B
env = initialize_distributed(seed=0, heap_bytes=3 << 30)


def lin(o, i, scale=0.05):
    return SimpleNamespace(weight=((torch.rand(o, i, device="cuda") * 2 - 1) * scale).to(BF), bias=None)


tree = SimpleNamespace(num_experts=E, top_k=topk, norm_topk_prob=True, gate=lin(E, H, 1.0),
                       experts=[SimpleNamespace(gate_proj=lin(I, H), up_proj=lin(I, H), down_proj=lin(H, I),
                                                act_fn=torch.nn.functional.silu) for _ in range(E)])
moe = TP_MoE(rank=0, world_size=1, env=env)
moe._init_parameters(tree)
moe._init_ctx(int(os.environ.get("CTX_M", "512")))
for T in [int(v) for v in os.environ.get("TOKENS", "1,4,16,17,32,64,128,256,512").split(",")]:
    xs = [(torch.rand(T, H, device="cuda") * 2 - 1).to(BF) for _ in range(8)]
    it = [0]
    def nat():
        i = it[0]; it[0] += 1
        return moe.dist_triton_fwd(xs[i % 8])
    def ref():
        i = it[0]; it[0] += 1
        return moe.torch_fwd_no_loop(xs[i % 8])
    if os.environ.get("NO_TORCH") == "1":
        rel = float("nan")
    else:
        a, b = moe.dist_triton_fwd(xs[0]).float(), moe.torch_fwd_no_loop(xs[0]).float()
        rel = (a - b).abs().max().item() / b.abs().max().item()
    t_n = graph_bench(nat, n=8)
    try:
        t_r = float("nan") if os.environ.get("NO_TORCH") == "1" else graph_bench(ref, n=8)
    except Exception as exc:  # noqa: BLE001
        t_r = float("nan")
    print(f"tokens {T}: native {t_n:.1f} us  torch (no loop) {t_r:.1f} us  rel diff {rel:.1e}", flush=True)
