"""The per-variant entry points of the reference (``triton_dist.kernels.nvidia.*``) run on the HIP backend: every
adapter lands on the native kernel and produces the collective's result on virtual ranks of one GPU."""
import pytest
import torch

from triton_distributed_gh_amd.utils.testing import VirtualCluster

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["fast_allgather_pull", "fast_allgather_push_2d", "fast_allgather_push_3d",
                                  "fast_allgather_push_2d_ll", "fast_allgather_push_numa_2d_ll"])
def test_low_latency_allgather_names(name):
    from triton_dist.kernels.nvidia import low_latency_allgather as lla

    w, shard = 4, (64, 130)
    vc = VirtualCluster(w, heap_bytes=1 << 26)
    try:
        ctxs = vc.collective(lambda r, env: lla.create_fast_allgather_context(r, max_buffer_size=shard[0] * shard[1] * 2, env=env))
        bufs = vc.collective(lambda r, env: env.heap.create_tensor([w, *shard], torch.bfloat16))
        vals = [torch.randn(shard, device="cuda").bfloat16() for _ in range(w)]
        for r in range(w):
            bufs[r][r].copy_(vals[r])
        torch.cuda.synchronize()
        vc.run(lambda r, env: getattr(lla, name)(ctxs[r], bufs[r]))
        for r in range(w):
            ctxs[r].check_error()
            for s in range(w):
                assert torch.equal(bufs[r][s], vals[s])
    finally:
        vc.close()


@pytest.mark.parametrize("name", ["allreduce_one_shot_push_intra_node", "allreduce_two_shot_push_intra_node",
                                  "allreduce_double_tree_intra_node", "allreduce_one_shot_multimem_intra_node"])
def test_allreduce_names(name):
    from triton_dist.kernels.nvidia import allreduce as ar

    w, n = 4, 8192 * 3 + 8
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ctxs = vc.collective(lambda r, env: ar.create_allreduce_ctx(8 << 20, r, w, env=env))
        xs = [(torch.rand(n, device="cuda") - 0.5).bfloat16() for _ in range(w)]
        torch.cuda.synchronize()
        outs = vc.run(lambda r, env: getattr(ar, name)(ctxs[r], xs[r]))
        ref = sum(x.float() for x in xs)
        for r in range(w):
            ctxs[r].check_error()
            assert torch.allclose(outs[r].float(), ref, atol=2e-2 * w, rtol=2e-2)
    finally:
        vc.close()


def test_reduce_scatter_and_copy_engine_names():
    from triton_dist.kernels.nvidia import allgather as ag
    from triton_dist.kernels.nvidia import reduce_scatter as rs

    w, M, N = 4, 512, 1024
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ctxs = vc.collective(lambda r, env: rs.create_reduce_scater_2d_ctx(M, N, r, w, w, torch.bfloat16, env=env))
        xs = [(torch.rand(M, N, device="cuda") - 0.5).bfloat16() for _ in range(w)]
        torch.cuda.synchronize()
        outs = vc.run(lambda r, env: rs.reduce_scatter_ring_push_1d_intra_node_ce(xs[r], ctxs[r]))
        ref = sum(x.float() for x in xs)
        for r in range(w):
            assert torch.allclose(outs[r].float(), ref[r * (M // w):(r + 1) * (M // w)], atol=2e-2 * w, rtol=2e-2)
        # a sum of row blocks (the reference's ring_reduce) and the stand-alone pad helper
        out = torch.empty(M // w, N, device="cuda", dtype=torch.bfloat16)
        rs.ring_reduce_tma(xs[0], out, 1, w)
        assert torch.allclose(out.float(), xs[0].float().view(w, M // w, N).sum(0), atol=4e-2, rtol=2e-2)

        # copy-engine all-gather, pull form: peers publish their slot, then every rank copies the others' in
        m = 64
        bufs = vc.collective(lambda r, env: env.heap.create_tensor([w * m, 96], torch.bfloat16))
        flags = vc.collective(lambda r, env: env.heap.create_tensor([w], torch.int32, arena="sig"))
        locs = [torch.randn(m, 96, device="cuda").bfloat16() for _ in range(w)]
        for r in range(w):
            bufs[r][r * m:(r + 1) * m].copy_(locs[r])
        torch.cuda.synchronize()
        vc.run(lambda r, env: ag.cp_engine_producer_all_gather_full_mesh_pull(
            r, w, locs[r], [env.heap.peer_view(bufs[r], p) for p in range(w)],
            [env.heap.peer_view(flags[r], p) for p in range(w)], torch.cuda.current_stream()))
        full = torch.cat(locs)
        for r in range(w):
            assert torch.equal(bufs[r], full) and flags[r].sum().item() == w - 1
    finally:
        vc.close()
