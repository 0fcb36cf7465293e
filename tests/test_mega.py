"""Mega-kernel runtime: graph / scheduler logic on the CPU, task kernels and the whole-model launch on
the GPU (scenarios of the reference's mega_triton_kernel/test/ops/test_{add,rms_norm,mlp_layer,
page_attn,allreduce}.py and test/models/test_qwen3.py)."""
import os

import pytest
import torch

os.environ["RANDOM_PARAMS"] = "1"
os.environ["TDIST_NO_HF_CONFIG"] = "1"

from triton_distributed_gh_amd.mega import (ModelBuilder, Qwen3Model, Task, TaskDependency, TaskType, check_schedule,  # noqa: E402
                                            round_robin_scheduler, zig_zag_scheduler)
from triton_distributed_gh_amd.mega.core import TASK_WORDS, encode_queues  # noqa: E402

BF = torch.bfloat16


def test_task_encoding_and_schedulers_cpu():
    t = Task(TaskType.LINEAR, 3, 7, [TaskDependency(2, 0, 4), TaskDependency(1, 7, 8)])
    enc = t.encoding()
    assert len(enc) == TASK_WORDS and enc[:10] == (1, 3, 7, 2, 2, 0, 4, 1, 7, 8)
    tasks = [Task(TaskType.ADD, 0, i, []) for i in range(10)]
    rr, zz = round_robin_scheduler(tasks, 4), zig_zag_scheduler(tasks, 4)
    assert [len(q) for q in rr] == [3, 3, 2, 2]
    assert [x.tile for x in zz[0]] == [0, 7, 8] and [x.tile for x in zz[3]] == [3, 4]
    buf, lens, mt = encode_queues(rr, "cpu")
    assert buf.shape == (4, 3, TASK_WORDS) and lens.tolist() == [3, 3, 2, 2] and mt == 3


def test_graph_dependencies_and_deadlock_check_cpu():
    b = ModelBuilder(num_queues=8)
    M, H, I = 4, 64, 128
    x, w1, xn = torch.zeros(M, H, dtype=BF), torch.ones(H, dtype=BF), torch.zeros(M, H, dtype=BF)
    wg, act = torch.zeros(2 * I, H, dtype=BF), torch.zeros(M, I, dtype=BF)
    wd, out = torch.zeros(H, I, dtype=BF), torch.zeros(M, H, dtype=BF)
    n = b.make_rms_norm(x, w1, xn)
    f1 = b.make_fc1(xn, wg, act)
    f2 = b.make_fc2(act, wd, out, residual=x)
    a = b.make_add(out, x, out)
    tasks = b.build_tasks()
    by_op = {}
    for t in tasks:
        by_op.setdefault(t.op, []).append(t)
    assert all(t.deps == [] for t in by_op[n.op_id])
    assert all(t.deps == [TaskDependency(n.op_id, 0, M)] for t in by_op[f1.op_id])          # whole norm
    assert all(t.deps == [TaskDependency(f1.op_id, 0, f1.num_tiles)] for t in by_op[f2.op_id])
    assert by_op[a.op_id][2].deps == [TaskDependency(f2.op_id, 0, f2.num_tiles)]
    for sched in (round_robin_scheduler, zig_zag_scheduler):
        check_schedule(sched(tasks, 8))
    # a consumer queued in front of its producer on the same queue can never start
    bad = [[by_op[f1.op_id][0], by_op[n.op_id][0]]] + [[t] for t in by_op[n.op_id][1:]]
    with pytest.raises(RuntimeError):
        check_schedule(bad)


def test_reused_tensors_get_write_after_read_dependencies_cpu():
    """One buffer written twice: the second writer waits for the reader of the first value unless a dependency it
    already has orders it behind that reader."""
    b = ModelBuilder(num_queues=8)
    M, H, N = 4, 64, 96
    x, r, g, xn = torch.zeros(M, H, dtype=BF), torch.zeros(M, H, dtype=BF), torch.ones(H, dtype=BF), torch.zeros(M, H, dtype=BF)
    w1, y1, hid = torch.zeros(N, H, dtype=BF), torch.zeros(M, N, dtype=BF), torch.zeros(M, H, dtype=BF)
    w2, y2 = torch.zeros(H, H, dtype=BF), torch.zeros(M, H, dtype=BF)
    n1 = b.make_rms_norm(x, g, xn)
    l1 = b.make_linear(xn, w1, y1)
    n2 = b.make_rms_norm(x, g, xn, 1e-6, residual=r, residual_out=hid)       # overwrites xn while l1 may still read it
    l2 = b.make_linear(xn, w2, y2)
    n3 = b.make_rms_norm(y2, g, xn)          # reads l2's output: l2 (the reader of xn) is already in front of it
    by_op = {}
    for t in b.build_tasks():
        by_op.setdefault(t.op, []).append(t)
    assert all(t.deps == [TaskDependency(l1.op_id, 0, l1.num_tiles)] for t in by_op[n2.op_id])
    assert all(t.deps == [TaskDependency(n2.op_id, 0, M)] for t in by_op[l2.op_id])
    assert all(t.deps == [TaskDependency(l2.op_id, 0, l2.num_tiles)] for t in by_op[n3.op_id])
    assert all(t.deps == [] for t in by_op[n1.op_id])


def _rel(a, b):
    return (a.float() - b.float()).abs().max().item() / max(b.float().abs().max().item(), 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,queues", [(1, 512, 256, 64), (5, 1536, 1024, 256), (16, 4096, 512, 32), (3, 200, 96, 16)])
def test_linear_task_modes(M, N, K, queues):
    torch.manual_seed(M + N)
    x = (torch.rand(M, K, device="cuda") - 0.5).to(BF)
    w = (torch.rand(N, K, device="cuda") - 0.5).to(BF)
    wgu = (torch.rand(2 * N, K, device="cuda") - 0.5).to(BF)
    res = (torch.rand(M, N, device="cuda") - 0.5).to(BF)
    o1, o2, o3 = (torch.empty(M, N, device="cuda", dtype=BF) for _ in range(3))
    b = ModelBuilder(num_queues=queues)
    b.make_linear(x, w, o1)
    b.make_linear(x, w, o2, residual=res)
    b.make_fc1(x, wgu, o3)
    for _ in range(2):
        b.run()
    torch.cuda.synchronize()
    b.check_error()
    ref = x.float() @ w.float().t()
    assert _rel(o1, ref) < 1e-2 and _rel(o2, ref + res.float()) < 1e-2
    gu = x.float() @ wgu.float().t()
    assert _rel(o3, torch.nn.functional.silu(gu[:, :N]) * gu[:, N:]) < 2e-2


@pytest.mark.gpu
def test_norm_add_silu_tasks():
    M, H = 7, 1024
    x, r = torch.randn(M, H, device="cuda").to(BF), torch.randn(M, H, device="cuda").to(BF)
    w = (1 + 0.1 * torch.randn(H, device="cuda")).to(BF)
    gu = torch.randn(M, 2 * H, device="cuda").to(BF)
    y1, y2, rout, s, sm = (torch.empty(M, H, device="cuda", dtype=BF) for _ in range(5))
    b = ModelBuilder(num_queues=16)
    b.make_rms_norm(x, w, y1, 1e-6)
    b.make_rms_norm(x, w, y2, 1e-6, residual=r, residual_out=rout)
    b.make_add(y1, y2, s)
    b.make_silu_mul_up(gu, sm)
    b.run()
    torch.cuda.synchronize()
    b.check_error()

    def rms(z):
        z = z.float()
        return z * torch.rsqrt(z.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float()

    xr = (x.float() + r.float()).to(BF)
    assert _rel(y1, rms(x)) < 1e-2 and _rel(y2, rms(xr)) < 1e-2 and torch.equal(rout, xr)
    assert _rel(s, y1.float() + y2.float()) < 1e-2
    assert _rel(sm, torch.nn.functional.silu(gu[:, :H].float()) * gu[:, H:].float()) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("B,Hq,Hkv,D,page,ctx_len", [(2, 8, 2, 128, 16, 300), (4, 4, 4, 64, 64, 1000), (1, 32, 8, 128, 1, 77)])
def test_qk_norm_rope_kv_and_paged_attention_tasks(B, Hq, Hkv, D, page, ctx_len):
    from triton_distributed_gh_amd.ops.elementwise import rope_cache
    from triton_distributed_gh_amd.ops.flash_decode import decode_reference

    torch.manual_seed(B)
    pages_per_seq = (ctx_len + 1 + page - 1) // page
    kc = torch.randn(B * pages_per_seq, page, Hkv, D, device="cuda").to(BF)
    vc = torch.randn(B * pages_per_seq, page, Hkv, D, device="cuda").to(BF)
    table = torch.arange(B * pages_per_seq, device="cuda", dtype=torch.int32).view(B, -1)
    lens = torch.full((B,), ctx_len, device="cuda", dtype=torch.int32)
    lens[0] = ctx_len // 2
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, device="cuda").to(BF)
    qn = (1 + 0.1 * torch.randn(D, device="cuda")).to(BF)
    kn = (1 + 0.1 * torch.randn(D, device="cuda")).to(BF)
    cos, sin = rope_cache(2048, D, 1e6, device="cuda")
    q = torch.empty(B, Hq, D, device="cuda", dtype=BF)
    out = torch.empty(B, Hq, D, device="cuda", dtype=BF)
    b = ModelBuilder(num_queues=64)
    b.make_qk_norm_rope_update_kvcache(qkv, qn, kn, cos, sin, lens, kc, vc, table, q)
    b.make_attn(q, kc, vc, table, lens, out)
    kc0, vc0 = kc.clone(), vc.clone()
    b.run()
    torch.cuda.synchronize()
    b.check_error()

    def norm_rope(z, wt, pos):
        z = z.float()
        z = (z * torch.rsqrt(z.pow(2).mean(-1, keepdim=True) + 1e-6) * wt.float()).to(BF).float()
        c, s = cos[pos].unsqueeze(1), sin[pos].unsqueeze(1)
        z1, z2 = z[..., : D // 2], z[..., D // 2:]
        return torch.cat([z1 * c - z2 * s, z2 * c + z1 * s], -1)

    qr = norm_rope(qkv[:, : Hq * D].view(B, Hq, D), qn, lens.long())
    kr = norm_rope(qkv[:, Hq * D:(Hq + Hkv) * D].view(B, Hkv, D), kn, lens.long())
    vr = qkv[:, (Hq + Hkv) * D:].view(B, Hkv, D)
    assert _rel(q, qr) < 1e-2
    for i in range(B):
        slot = int(table[i, int(lens[i]) // page]) * page + int(lens[i]) % page
        assert _rel(kc.view(-1, Hkv, D)[slot], kr[i]) < 1e-2 and torch.equal(vc.view(-1, Hkv, D)[slot], vr[i])
    kc0.view(-1, Hkv, D)[[int(table[i, int(lens[i]) // page]) * page + int(lens[i]) % page for i in range(B)]] = kr.to(BF)
    vc0.view(-1, Hkv, D)[[int(table[i, int(lens[i]) // page]) * page + int(lens[i]) % page for i in range(B)]] = vr
    ref = decode_reference(q, kc0, vc0, lens + 1, table, D ** -0.5)
    assert _rel(out, ref) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("w", [2, 4])
def test_allreduce_and_barrier_tasks_on_virtual_ranks(w):
    from triton_distributed_gh_amd.utils.testing import VirtualCluster

    M, H, K = 4, 1024, 256
    vc = VirtualCluster(w, heap_bytes=1 << 24)
    try:
        builders, outs, xs, ws = [], [], [], []
        for r, env in enumerate(vc.envs):
            b = ModelBuilder(r, w, w, num_queues=16, env=env)
            x = (torch.rand(M, K, device="cuda") - 0.5).to(BF)
            wt = (torch.rand(H, K, device="cuda") - 0.5).to(BF)
            part = b.create_symm_tensor((M, H), BF)
            out = torch.empty(M, H, device="cuda", dtype=BF)
            b.make_linear(x, wt, part)
            b.make_allreduce(part, out)
            b.make_barrier_all_intra_node()
            builders.append(b), outs.append(out), xs.append(x), ws.append(wt)
        vc.collective(lambda r, env: builders[r].compile())
        for it in range(3):
            vc.run(lambda r, env: builders[r].run())
            ref = sum((xs[r].float() @ ws[r].float().t()).to(BF).float() for r in range(w))
            for r in range(w):
                builders[r].check_error()
                assert _rel(outs[r], ref) < 2e-2
    finally:
        vc.close()


@pytest.mark.gpu
def test_graph_backend_matches_the_interpreter_on_a_small_graph():
    """The same recorded ops through both back-ends: norm -> linear and norm(+residual) -> fc1 are fused launches in
    the graph form (up to the row count the dispatch fuses at), the rest lowers one to one; outputs land in the
    recorded tensors either way."""
    M, H, N, I = 2, 1024, 768, 512
    torch.manual_seed(3)
    x, r = torch.randn(M, H, device="cuda").to(BF), torch.randn(M, H, device="cuda").to(BF)
    nw = (1 + 0.1 * torch.randn(H, device="cuda")).to(BF)
    w1 = (torch.randn(N, H, device="cuda") / 32).to(BF)
    wgu = (torch.randn(2 * I, H, device="cuda") / 32).to(BF)
    w2 = (torch.randn(H, I, device="cuda") / 32).to(BF)
    results = {}
    for backend in ("interpreter", "graph"):
        new = lambda *s: torch.zeros(*s, device="cuda", dtype=BF)   # noqa: E731
        xn, y1, hid, act, y2, s_ = new(M, H), new(M, N), new(M, H), new(M, I), new(M, H), new(M, H)
        b = ModelBuilder(num_queues=32)
        b.make_rms_norm(x, nw, xn, 1e-6)
        b.make_linear(xn, w1, y1)
        b.make_rms_norm(x, nw, xn, 1e-6, residual=r, residual_out=hid)
        b.make_fc1(xn, wgu, act)
        b.make_fc2(act, w2, y2, residual=hid)
        b.make_add(y2, hid, s_)
        c = b.compile(backend=backend)
        assert c["backend"] == backend
        for _ in range(3):                      # graph form: eager, capture + replay, replay
            b.run()
        torch.cuda.synchronize()
        b.check_error()
        results[backend] = (y1, hid, s_)
        if backend == "graph":
            assert len(c["program"].steps) == 4 and c["program"].graph is not None, c["program"].launch_names
    for a, g_ in zip(results["interpreter"], results["graph"]):
        assert _rel(g_, a.float()) < 2e-2
    assert torch.equal(results["interpreter"][1], results["graph"][1])      # the residual stream is exact in both


@pytest.mark.gpu
@pytest.mark.parametrize("w,name,backend", [(1, "tdist/tiny-dense", "interpreter"), (2, "tdist/tiny-dense", "interpreter"),
                                            (1, "tdist/tiny-dense-d128", "graph"), (2, "tdist/tiny-dense-d128", "graph"),
                                            (1, "tdist/tiny-dense", "auto")])
def test_qwen3_mega_forward_matches_torch_decode(w, name, backend):
    from triton_distributed_gh_amd.models import AutoLLM, KV_Cache, ModelConfig
    from triton_distributed_gh_amd.utils.testing import VirtualCluster

    B, prompt, L = 4, 24, 256
    torch.manual_seed(7)
    g = AutoLLM.from_pretrained(ModelConfig(model_name=name, max_length=L, dtype=BF, rank=0, world_size=1))
    gkv = KV_Cache(g.num_layers, B, L, g.num_key_value_heads, g.head_dim, BF, 1)
    ids = torch.randint(0, 1000, (B, prompt), device="cuda")
    pos = torch.arange(prompt, device="cuda")[None].repeat(B, 1)
    logits = g.inference(ids, pos, gkv)
    gkv.kv_offset.fill_(prompt)
    vc = VirtualCluster(w, heap_bytes=1 << 26)
    try:
        models = [AutoLLM.from_pretrained(ModelConfig(model_name=name, max_length=L, dtype=BF, rank=r, world_size=w),
                                          env=vc.envs[r]) for r in range(w)]
        megas = vc.collective(lambda r, env: Qwen3Model(models[r], B, L, page_size=16, num_queues=64 // w, env=env,
                                                        backend=backend))
        # head_dim 64 has no fused attention launch: "auto" falls back to the interpreter
        assert megas[0].backend == ("graph" if backend == "graph" else "interpreter")
        hl = g.num_key_value_heads // w
        for r, m in enumerate(megas):       # copy the golden prefill cache into the paged caches
            for li in range(g.num_layers):
                for b_ in range(B):
                    kc = m.kv_cache.k_cache[li].view(-1, hl, g.head_dim)
                    vcache = m.kv_cache.v_cache[li].view(-1, hl, g.head_dim)
                    base = int(m.kv_cache.block_table[b_, 0]) * 16
                    kc[base:base + prompt] = gkv.k_cache[li, b_, :prompt, r * hl:(r + 1) * hl]
                    vcache[base:base + prompt] = gkv.v_cache[li, b_, :prompt, r * hl:(r + 1) * hl]
            m.set_kv_len(prompt)
        tok = logits[:, -1].argmax(-1, keepdim=True)
        for step in range(4):
            ref = g.inference(tok, gkv.get_kv_len()[:, None].long(), gkv)
            gkv.inc_offset()
            torch.cuda.synchronize()
            outs = vc.run(lambda r, env: megas[r].mega_forward(tok))
            for r in range(w):
                megas[r].check_error()
                assert _rel(outs[r], ref) < 6e-2, (step, r)
            tok = ref[:, -1].argmax(-1, keepdim=True)
    finally:
        vc.close()


def test_task_kinds_configs_and_facade_cpu():
    """Every registered op type has a config class and names the ModelBuilder method and interpreter body that
    serve it; the reference's module paths for tasks / configs / layer builders resolve."""
    import importlib

    from triton_distributed_gh_amd.mega import tasks
    from triton_distributed_gh_amd.mega.core import registered_task_types
    from triton_distributed_gh_amd.mega.model_builder import ModelBuilder as MB

    assert set(tasks.TASK_KINDS) == set(registered_task_types())
    for name, kind in tasks.TASK_KINDS.items():
        assert hasattr(MB, kind.builder_method) and issubclass(kind.config_cls, tasks.ConfigBase), name
        assert isinstance(tasks.default_config(name).to_dict(), dict)
    assert tasks.linear_block_n(tasks.LinearConfig(BLOCK_SIZE_N=32)) == 32 and tasks.linear_block_n(None) == 0
    with pytest.raises(ValueError):
        tasks.linear_block_n(tasks.LinearConfig(BLOCK_SIZE_N=48))
    for mod, attr in [("triton_dist.mega_triton_kernel.tasks.linear", "LinearConfig"),
                      ("triton_dist.mega_triton_kernel.core.config", "ConfigBase"),
                      ("triton_dist.mega_triton_kernel.models.layers.tp_mlp", "TPMLPBuilder"),
                      ("triton_dist.mega_triton_kernel.models.layers.tp_attn", "TPAttnBuilder"),
                      ("triton_dist.mega_triton_kernel.models.utils", "prepare_cos_sin_cache"),
                      ("triton_dist.mega_triton_kernel.kernels", "TASK_BODIES")]:
        assert hasattr(importlib.import_module(mod), attr), (mod, attr)


@pytest.mark.gpu
@pytest.mark.parametrize("bn", [16, 32, 64])
def test_linear_config_block_n_and_mlp_builder(bn):
    """A LinearConfig fixes the columns per task; TPMLPBuilder shards a plain MLP module and records the block."""
    from triton_distributed_gh_amd.mega import LinearConfig, TPMLPBuilder
    from triton_distributed_gh_amd.mega.kernels import check_abi

    check_abi()
    torch.manual_seed(bn)
    H, I, M = 512, 1024, 3
    mlp = torch.nn.Module()
    mlp.gate_proj, mlp.up_proj = torch.nn.Linear(H, I, bias=False), torch.nn.Linear(H, I, bias=False)
    mlp.down_proj = torch.nn.Linear(I, H, bias=False)
    mlp = mlp.to("cuda", BF)
    x = (torch.rand(M, H, device="cuda") - 0.5).to(BF)
    b = ModelBuilder(num_queues=64)
    blk = TPMLPBuilder(b, rank=0, world_size=1)._init_parameters(mlp)
    cfg = LinearConfig(BLOCK_SIZE_N=bn)
    out = blk.build_fwd(x, fc1_config=cfg, fc2_config=cfg)
    assert [op.num_tiles for op in b.ops] == [I // bn, H // bn]
    b.run()
    torch.cuda.synchronize()
    b.check_error()
    h = torch.nn.functional.silu(x.float() @ mlp.gate_proj.weight.float().t()) * (x.float() @ mlp.up_proj.weight.float().t())
    ref = h.to(BF).float() @ mlp.down_proj.weight.float().t()
    assert _rel(out, ref) < 2e-2
