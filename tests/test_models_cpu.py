"""Model layer on the CPU backend: tiny dense / MoE models, engine decode loop, every forward mode
against the single-rank torch forward (scenario of the reference's test_tp_e2e.py --check and
test_e2e_inference.py, python/triton_dist/test/amd/test_tp_e2e.py:80-91)."""
import os

import pytest
import torch

os.environ["RANDOM_PARAMS"] = "1"
os.environ["TDIST_NO_HF_CONFIG"] = "1"

from triton_distributed_gh_amd.models import AutoLLM, Engine, KV_Cache, ModelConfig  # noqa: E402
from triton_distributed_gh_amd.models.utils import sample_token  # noqa: E402
from triton_distributed_gh_amd.utils.testing import VirtualCluster  # noqa: E402


def _golden(name, ids, pos, dtype=torch.float32, steps=2):
    m = AutoLLM.from_pretrained(ModelConfig(model_name=name, max_length=64, dtype=dtype, rank=0, world_size=1))
    kv = KV_Cache(m.num_layers, ids.shape[0], 64, m.num_key_value_heads, m.head_dim, dtype, 1, device="cpu")
    outs = [m.inference(ids, pos, kv)]
    kv.kv_offset.fill_(ids.shape[1])
    for _ in range(steps):
        tok = outs[-1][:, -1].argmax(-1, keepdim=True)
        outs.append(m.inference(tok, kv.get_kv_len()[:, None].long(), kv))
        kv.inc_offset()
    return outs


@pytest.mark.parametrize("name,mode", [("tdist/tiny-dense", "triton_dist"), ("tdist/tiny-dense", "triton_dist_AR"),
                                       ("tdist/tiny-dense", "triton_dist_gemm_ar"), ("tdist/tiny-moe", "triton_dist")])
def test_tp2_modes_match_single_rank(name, mode):
    w, bsz, prompt = 2, 4, 6
    torch.manual_seed(0)
    ids = torch.randint(0, 1000, (bsz, prompt))
    pos = torch.arange(prompt)[None].repeat(bsz, 1)
    gold = _golden(name, ids, pos)
    vc = VirtualCluster(w, heap_bytes=1 << 25, backend="host")
    models = [AutoLLM.from_pretrained(ModelConfig(model_name=name, max_length=64, dtype=torch.float32, rank=r,
                                                  world_size=w), env=vc.envs[r]) for r in range(w)]
    kvs = [KV_Cache(models[r].num_layers, bsz, 64, models[r].num_key_value_heads, models[r].head_dim, torch.float32, w,
                    device="cpu") for r in range(w)]
    for r in range(w):
        models[r].set_fwd(mode)
    if mode == "triton_dist":
        vc.collective(lambda r, env: models[r].init_triton_dist_ctx(max_M=bsz * prompt))
    elif mode == "triton_dist_AR":
        vc.collective(lambda r, env: models[r].init_triton_dist_AR_ctx(max_M=bsz * prompt))
    else:
        vc.collective(lambda r, env: models[r].init_triton_dist_gemm_ar_ctx(max_M=bsz * prompt))
    sharded = mode == "triton_dist"
    per = bsz // w

    def shard(x, r):
        return x[r * per:(r + 1) * per] if sharded else x

    def gather(outs):
        return torch.cat(outs) if sharded else outs[0]

    outs = vc.run(lambda r, env: models[r].inference(shard(ids, r), pos, kvs[r]))
    assert torch.allclose(gather(outs), gold[0], atol=2e-3, rtol=2e-3)
    for kv in kvs:
        kv.kv_offset.fill_(prompt)
    for step in range(2):
        tok = gold[step][:, -1].argmax(-1, keepdim=True)
        outs = vc.run(lambda r, env: models[r].inference(shard(tok, r), kvs[r].get_kv_len()[:, None].long(), kvs[r]))
        assert torch.allclose(gather(outs), gold[step + 1], atol=2e-3, rtol=2e-3)
        for kv in kvs:
            kv.inc_offset()
    vc.close()


def test_engine_generates_on_cpu():
    eng = Engine(ModelConfig(model_name="tdist/tiny-dense", max_length=64, dtype=torch.float32, rank=0, world_size=1))
    out = eng.serve(torch.randint(0, 1000, (2, 5)), gen_len=3)
    assert out.shape == (2, 3) and eng.last_decode_seconds is not None
    out2 = eng.serve(torch.randint(0, 1000, (1, 4)), gen_len=2)      # the engine is reusable
    assert out2.shape == (1, 2)


def test_sample_token_top_p_and_argmax():
    logits = torch.tensor([[0.0, 10.0, 0.0, 0.0], [5.0, 0.0, 0.0, 0.0]])
    assert sample_token(logits, temperature=0.0).tolist() == [[1], [0]]
    torch.manual_seed(0)
    for _ in range(10):
        assert sample_token(logits, temperature=0.5, top_p=0.5).tolist() == [[1], [0]]
    tok = sample_token(torch.zeros(3, 16), temperature=1.0, top_p=1.0, top_k=4)
    assert tok.shape == (3, 1) and int(tok.max()) < 16


def test_moe_golden_without_expert_loop_matches_the_loop():
    """TP_MoE.torch_fwd_no_loop (gathered expert weights, one batched product, no host synchronisation - the form a
    decode step can capture) against the loop over experts, fp32 on the CPU."""
    from types import SimpleNamespace

    from triton_distributed_gh_amd.layers import TP_MoE

    torch.manual_seed(3)
    H, I, E, topk, T = 64, 96, 8, 2, 10
    lin = lambda o, i: SimpleNamespace(weight=(torch.rand(o, i) * 2 - 1) * 0.2, bias=None)   # noqa: E731
    tree = SimpleNamespace(num_experts=E, top_k=topk, norm_topk_prob=True, gate=lin(E, H),
                           experts=[SimpleNamespace(gate_proj=lin(I, H), up_proj=lin(I, H), down_proj=lin(H, I),
                                                    act_fn=torch.nn.functional.silu) for _ in range(E)])
    moe = TP_MoE(rank=0, world_size=1)
    moe._init_parameters(tree)
    x = torch.rand(2, T // 2, H) * 2 - 1
    ref = moe.torch_fwd(x)
    got = moe.torch_fwd_no_loop(x)
    assert got.shape == x.shape
    torch.testing.assert_close(got, ref, atol=1e-5, rtol=1e-4)
