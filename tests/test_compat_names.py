"""The reference's host-level names resolve on the facade and the adapters behind them do what they say (CPU)."""
import importlib.util
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/python/triton_dist")

# Names with no counterpart: they generate or describe Triton compiler output (AST sources, C launchers, pybind
# glue); kernels here are HIP sources compiled by hipcc and bound through ctypes, so there is nothing to generate.
NO_COUNTERPART = {
    "triton_dist.tools.compile.compile": None,           # whole module: Triton AST -> C source
    "triton_dist.tools.compile_aot": {"make_global_decl_with_algo_info", "map_arg_to_pybind_arg", "map_ty_to_pybind_ty",
                                      "gen_pybind_args_with_full_args", "gen_pybind_signature_with_full_args",
                                      "make_global_pybind_with_algo_info", "make_algo_info_condition",
                                      "make_kernel_with_algo_info_param", "make_func_pointers_impl",
                                      "make_kernel_load_def_impl"},
    # in-module self tests / perf scripts of the reference
    "triton_dist.kernels.nvidia.threadblock_swizzle_ag_moe_triton": {
        "get_default_profile_context", "test_threadblock_swizzle_ag_moe_naive", "test_threadblock_swizzle_ag_moe_random",
        "perf_threadblock_swizzle_ag_moe_random"},
}


def _gap_module():
    spec = importlib.util.spec_from_file_location("facade_gap", ROOT / "scripts" / "dev" / "facade_gap.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_reference_host_name_resolves():
    if not REF.exists():
        pytest.skip("reference tree not present")
    sys.path.insert(0, str(ROOT))
    left = {}
    for mod, names in _gap_module().gaps(REF).items():
        allowed = NO_COUNTERPART.get(mod, set())
        rest = [] if allowed is None else [n for n in names if n not in allowed]
        if rest:
            left[mod] = rest
    assert not left, left


def test_double_tree_is_two_spanning_trees():
    from triton_dist.kernels.nvidia.allreduce import get_tree_parent_and_children

    for N in (2, 4, 8, 16):
        for which in (0, 1):
            info = [get_tree_parent_and_children(N, r)[which] for r in range(N)]
            roots = [r for r in range(N) if info[r][0] < 0]
            assert len(roots) == 1
            for r in range(N):
                p, c0, c1 = info[r]
                if p >= 0:
                    assert r in info[p][1:], (N, which, r, info)
                for c in (c0, c1):
                    if c >= 0:
                        assert info[c][0] == r
        if N >= 4:
            # a leaf of one tree is an inner node of the other (ranks 0 and N-1 excepted: leaves in both / chain ends)
            leaves0 = {r for r in range(N) if get_tree_parent_and_children(N, r)[0][1:] == (-1, -1)}
            leaves1 = {r for r in range(N) if get_tree_parent_and_children(N, r)[1][1:] == (-1, -1)}
            assert len(leaves0 & leaves1) <= 1


def test_swizzle_model_matches_native_align_order():
    from triton_dist.kernels.nvidia import threadblock_swizzle_ag_moe as sw
    from triton_distributed_gh_amd.ops.moe import moe_align_reference

    w, E, topk, tpr = 4, 6, 2, 384
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, E, (w * tpr, topk), generator=g, dtype=torch.int32)
    src = torch.arange(w * tpr).repeat_interleave(topk) // tpr
    counts = [[int(((ids.reshape(-1) == e) & (src == r)).sum()) for e in range(E)] for r in range(w)]
    sw.check_with_token_cnt_per_rank_per_expert(counts)
    for rank in range(w):
        idx = moe_align_reference(ids, E, w, rank, tpr)
        nb = int(idx.meta[0])
        tiles = sw.swizzled_tiles(rank, w, 256, counts)
        assert len(tiles) == nb
        # native block index b enumerates experts, then tiles: rebuild it from the model's (expert, tile)
        per_e = [(c + 255) // 256 for c in (sum(col) for col in zip(*counts))]
        base = [sum(per_e[:e]) for e in range(E)]
        # the native order walks eight contiguous ranges of blocks (one per XCD), each in the model's arrival order
        model = [base[t.expert_id] + t.tiled_m for t in tiles]
        want = [b for q in range(8) for b in model if q * nb // 8 <= b < (q + 1) * nb // 8]
        assert want == idx.order[:nb].tolist()
        for t in tiles:
            b = base[t.expert_id] + t.tiled_m
            assert [t.segment_start, t.segment_end] == idx.seg_range[b].tolist()
    stage, e, tile = sw.threadblock_swizzle_ag_moe(0, 1, E, w, 256, counts)
    assert stage == tile.stage and e == tile.expert_id
    for gen in (sw.generate_token_cnts_per_rank_per_expert_uniform, sw.generate_token_cnts_per_rank_per_expert_random,
                sw.generate_token_cnts_per_rank_per_expert_random_with_many_zeros):
        sw.check_with_token_cnt_per_rank_per_expert(gen(1024, 8, 4))


def test_gemm_rs_swizzle_visits_next_rank_first_and_own_rows_last():
    from triton_dist.kernels.nvidia.gemm_rs_threadblock_swizzle import (check_gemm_rs_swizzle,
                                                                       threadblock_swizzle_gemm_reduce_scatter as f)

    for M, w, nn, bm in ((4096, 8, 1, 256), (4096, 8, 2, 128), (1536, 4, 1, 256), (8 * 200, 8, 1, 128)):
        check_gemm_rs_swizzle(M, w, nn, bm)
    assert f(0, 4096, 2, 8, 1, 256) == 3 * 2 and f(15, 4096, 2, 8, 1, 256) == 2 * 2 + 1

    from triton_dist.kernels.nvidia.ag_gemm_threadblock_swizzle import check_ag_gemm_swizzle, threadblock_swizzle_allgather_gemm as g

    for M, w, nn, bm in ((4096, 8, 1, 256), (4096, 8, 2, 128), (1536, 4, 1, 256), (8 * 200, 8, 1, 128)):
        check_ag_gemm_swizzle(M, w, nn, bm)
    assert [g(i, 2048, 2, 4, 1, 256) for i in range(8)] == [4, 5, 6, 7, 0, 1, 2, 3]       # own rows, then ranks 3, 0, 1


def test_overload_dispatch_reaches_native_symbols():
    import ctypes

    from triton_dist.language.core import dispatch, extern_elementwise

    table = {(): "tdh_registry_num_kernels", (torch.float32, int): ("no_such_symbol", ctypes.c_int)}
    assert dispatch(None, "host", "", [], table) >= 0
    assert dispatch(lambda sym, args: (sym, len(args)), "host", "", [], table) == ("tdh_registry_num_kernels", 0)
    with pytest.raises(ValueError):
        dispatch(None, "host", "", [1.5], table)
    with pytest.raises(AttributeError):
        extern_elementwise("host", "", [torch.zeros(2), 3], table)


def test_moe_index_helpers():
    from triton_dist.kernels.nvidia import moe_utils as mu

    ids = torch.tensor([[2, 0], [1, 2], [0, 1], [2, 2]], dtype=torch.int32)
    scatter = mu.calc_scatter_index_torch(ids, 3)
    gather = mu.calc_gather_index_torch(ids, 3)
    assert sorted(scatter.tolist()) == list(range(8))
    assert mu.calc_gather_index_from_scatter_index(scatter).tolist() == gather.tolist()
    flat = ids.reshape(-1)
    assert flat[gather.long()].tolist() == sorted(flat.tolist())
    assert mu.histogram_by_expert_torch(ids, 3).tolist() == [2, 2, 4]
    x = torch.arange(8, dtype=torch.float32).view(8, 1)
    assert mu.reduce_topk_non_tma(x, 2).view(-1).tolist() == [1.0, 5.0, 9.0, 13.0]

    from triton_dist.kernels.nvidia.gemm_reduce_scatter import padded_to_BLOCK_M

    p = padded_to_BLOCK_M(torch.ones(6, 2), 2, 4)
    assert p.shape == (8, 2) and p[3].sum() == 0 and p[4].sum() == 2


def test_dispatch_send_runs():
    from triton_dist.kernels.nvidia.ep_a2a import get_dispatch_send_reqs_for_target_node

    node = torch.tensor([[0, 1], [1, 1], [0, 0], [0, 0], [1, 0], [0, 0]])
    st, ed = get_dispatch_send_reqs_for_target_node(node, 1)
    assert st[:2].tolist() == [0, 4] and ed[:2].tolist() == [2, 5] and st.dtype == torch.int32 and st.numel() == 6


def test_perf_model_names():
    from triton_dist.kernels.nvidia import comm_perf_model as cm
    from triton_dist.kernels.nvidia import gemm_perf_model as gm

    assert gm.is_fp8_dtype(torch.float8_e4m3fn) and not gm.is_fp8_dtype(torch.bfloat16)
    bf16 = gm.get_tensorcore_tflops_by_calc(torch.bfloat16)
    assert 2000 < bf16 < 3000 and gm.get_tensorcore_tflops_by_calc(torch.float8_e4m3fn) == 2 * bf16
    assert gm.get_tensorcore_tflops(torch.bfloat16) > 1000 and gm.get_dram_gbps_by_device_name("MI355X") > 5000
    assert torch.bfloat16 in gm.get_tensorcore_dtype_support()
    assert isinstance(cm.get_network_interfaces(), list) and cm.get_max_nic_bandwidth_gpbs() >= 0


def test_adapters_on_a_virtual_cluster():
    from triton_dist.kernels.nvidia import allgather as ag
    from triton_dist.kernels.nvidia.ep_a2a import get_ag_splits_and_recv_offset_for_dispatch
    from triton_dist.language.extra import libshmem_device as shm
    from triton_distributed_gh_amd.utils.testing import VirtualCluster

    w, epr, topk = 2, 2, 1
    vc = VirtualCluster(w, heap_bytes=1 << 20, backend="host")
    bufs = vc.collective(lambda r, env: env.heap.create_tensor([w * 4, 3], torch.float32))
    splits = vc.collective(lambda r, env: env.heap.create_tensor([w, w * epr], torch.int32))
    coll = vc.collective(lambda r, env: env.heap.create_tensor([w, 4], torch.float32))
    from triton_distributed_gh_amd.ops.common_ops import BarrierAllContext

    vc.collective(lambda r, env: setattr(env, "_barrier_all_ctx", BarrierAllContext(env=env)))
    mine = [[3, 1, 0, 2], [1, 1, 4, 0]]

    def fn(r, env):
        local = torch.full((4, 3), float(r + 1))
        bufs[r][r * 4:(r + 1) * 4].copy_(local)                      # publish own slot, then pull the peers'
        shm.barrier_all()
        ag.cp_engine_producer_all_gather_full_mesh_pull(r, w, local, [env.heap.peer_view(bufs[r], p) for p in range(w)], None)
        offs, recv, held = get_ag_splits_and_recv_offset_for_dispatch(torch.tensor(mine[r], dtype=torch.int32), splits[r],
                                                                      None, topk, w, epr)
        src = torch.full((4, ), float(10 + r))
        shm.barrier_all()
        shm.fcollectmem(0, coll[r], _symm_src(env, coll, r, src), 16)
        return bufs[r][:, 0].tolist(), offs.tolist(), recv.tolist(), held.tolist(), coll[r][:, 0].tolist()

    stage = vc.collective(lambda r, env: env.heap.create_tensor([4], torch.float32))

    def _symm_src(env, coll, r, src):
        stage[r].copy_(src)
        shm.barrier_all()
        return stage[r]

    out = vc.run(fn)
    for r in range(w):
        rows, offs, recv, held, gathered = out[r]
        assert rows == [1.0] * 4 + [2.0] * 4
        # owner 0 holds experts 0,1: sources send (3,1) and (1,1) -> rows laid out e0:[src0 3][src1 1] e1:[src0 1][src1 1]
        assert offs[0] == [[0, 3], [4, 5]] and offs[1] == [[0, 0], [4, 6]]
        assert recv == [6, 6] and held == [6, 6]
        assert gathered == [10.0, 11.0]
    vc.close()


def test_mega_task_api_builds_and_schedules():
    from triton_dist.mega_triton_kernel.core.builder import TaskBuilderBase
    from triton_dist.mega_triton_kernel.core.registry import registry
    from triton_dist.mega_triton_kernel.core.scheduler import SchedulingStrategy, enque_tasks
    from triton_dist.mega_triton_kernel.core.task_base import CodeGenKey, DeviceProp, TaskBase, TaskIDManager
    from triton_dist.mega_triton_kernel.kernels.task_context import TaskBaseInfo
    from triton_dist.mega_triton_kernel.tasks.linear import LinearTask, LinearTaskBuilder, codegen_linear, linear_config_factory
    from triton_dist.mega_triton_kernel.tasks.norm import RMSNormTaskBuilder
    from triton_distributed_gh_amd.mega import ModelBuilder

    assert len(registry.op_types()) == 14 and registry.get_op_mapping("linear") is LinearTask
    assert registry.get_builder(LinearTask) is LinearTaskBuilder and issubclass(LinearTaskBuilder, TaskBuilderBase)
    assert codegen_linear() == "task_linear" and issubclass(LinearTask, TaskBase)
    assert linear_config_factory(BLOCK_SIZE_N=32, unknown=1).BLOCK_SIZE_N == 32
    assert CodeGenKey(1) == LinearTask.get_codegen_key() and LinearTask.get_codegen_key().only_use_task_type()
    with pytest.raises(ValueError):
        CodeGenKey(1, layer_id=3)
    with pytest.raises(ValueError):
        registry.get_op_mapping("conv")

    TaskIDManager.reset_task_ids()
    b = ModelBuilder(num_queues=8)
    x, wgt, y = torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(128, 64, dtype=torch.bfloat16), torch.zeros(4, 128, dtype=torch.bfloat16)
    nw, yn = torch.ones(128, dtype=torch.bfloat16), torch.zeros(4, 128, dtype=torch.bfloat16)
    lin = LinearTaskBuilder.build_tasks(DeviceProp(8), 0, None, [[x, wgt], [y]], {"config": linear_config_factory(BLOCK_SIZE_N=32)},
                                        builder=b)
    assert len(lin) == 4 and [t.tile_id_or_start for t in lin] == [0, 1, 2, 3] and lin[0].task_id == 0
    from triton_distributed_gh_amd.mega.core import TaskDependency

    norm = RMSNormTaskBuilder.build_tasks(DeviceProp(8), 0, TaskDependency(lin[0].op_id, 0, 4), [[y, nw], [yn]], {"eps": 1e-6},
                                          builder=b)
    assert len(norm) == 4 and norm[0].task_id == 1 and norm[0].op_id == 1
    q, lens = enque_tasks(8, lin + norm, SchedulingStrategy.ZIG_ZAG, device="cpu")
    assert q.shape == (8, 1, 16) and lens.tolist() == [1] * 8 and SchedulingStrategy.ZIG_ZAG == "zig_zag"
    rec = TaskBaseInfo.decode(norm[2].encoding())
    assert (rec.op, rec.tile, rec.deps[0].hi) == (1, 2, 4)
    with pytest.raises(RuntimeError):           # a consumer queued ahead of its producer on the same queue deadlocks
        enque_tasks(1, norm + lin, device="cpu")


def test_qwen3_layer_builder_records_a_layer():
    from triton_dist.mega_triton_kernel.models.qwen3 import Qwen3LayerBuilder, shard_local
    from triton_distributed_gh_amd.mega import ModelBuilder, prepare_cos_sin_cache
    from triton_distributed_gh_amd.mega.core import check_schedule, round_robin_scheduler
    from triton_distributed_gh_amd.models.kv_cache import PagedKVCache

    H, D, Hq, Hkv, inter, B = 128, 64, 2, 1, 256, 2

    class Norm(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            self.weight, self.variance_epsilon = torch.nn.Parameter(torch.ones(n, dtype=torch.bfloat16)), 1e-5

    lin = lambda o, i: torch.nn.Linear(i, o, bias=False, dtype=torch.bfloat16)  # noqa: E731
    layer = torch.nn.Module()
    layer.self_attn, layer.mlp = torch.nn.Module(), torch.nn.Module()
    layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj = lin(Hq * D, H), lin(Hkv * D, H), lin(Hkv * D, H)
    layer.self_attn.o_proj, layer.self_attn.q_norm, layer.self_attn.k_norm = lin(H, Hq * D), Norm(D), Norm(D)
    layer.mlp.gate_proj, layer.mlp.up_proj, layer.mlp.down_proj = lin(inter, H), lin(inter, H), lin(H, inter)
    layer.input_layernorm, layer.post_attention_layernorm = Norm(H), Norm(H)
    assert shard_local(torch.arange(8).view(4, 2), 2, 0, 1).tolist() == [[4, 5], [6, 7]]

    b = ModelBuilder(num_queues=16)
    lb = Qwen3LayerBuilder(b, 0, D).init_parameters(layer)
    cos, sin = prepare_cos_sin_cache(D, 128, 1e4)
    kv = PagedKVCache(1, B, 128, Hkv, D, page_size=64, dtype=torch.bfloat16, world_size=1, device="cpu")
    hidden = torch.zeros(B, 1, H, dtype=torch.bfloat16)
    assert lb.build_fwd(hidden, cos, sin, kv) is hidden
    names = [op.name for op in b.ops]
    assert names[0] == "l0.norm1" and names[-1] == "l0.residual" and "l0.attn_split" in names and "l0.fc1" in names
    check_schedule(round_robin_scheduler(b.build_tasks(), 16))
    assert lb.post_norm_eps == 1e-5


@pytest.mark.parametrize("zigzag", [True, False])
def test_copy_engine_kv_all_gather_places_every_shard(zigzag):
    """Every rank's K / V shard rows land at their place of the full per-sequence K / V of every peer
    (zig-zag: half-shards at slabs r and 2w-1-r of 2w), checked against a directly built full tensor."""
    from triton_dist.kernels.nvidia.sp_ag_attention_intra_node import cp_engine_producer_kv_all_gather

    w, H, D = 2, 2, 8
    full_lens = [16, 32]
    cu = [0, 16, 48]
    full_k = [torch.randn(L, H, D) for L in full_lens]
    full_v = [torch.randn(L, H, D) for L in full_lens]

    def shard(full, r):
        rows = []
        for t in full:
            L = t.shape[0]
            if zigzag:
                h = L // (2 * w)
                rows += [t[r * h:(r + 1) * h], t[(2 * w - 1 - r) * h:(2 * w - r) * h]]
            else:
                s = L // w
                rows.append(t[r * s:(r + 1) * s])
        return torch.cat(rows)

    kb = [torch.zeros(sum(full_lens), H, D) for _ in range(w)]
    vb = [torch.zeros(sum(full_lens), H, D) for _ in range(w)]
    for r in range(w):
        cp_engine_producer_kv_all_gather(shard(full_k, r), shard(full_v, r), kb[r], vb[r], kb, vb, torch.tensor(cu), r, w,
                                         enable_zig_zag=zigzag)
    for r in range(w):
        assert torch.equal(kb[r], torch.cat(full_k)) and torch.equal(vb[r], torch.cat(full_v))
