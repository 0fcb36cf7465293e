"""The tile family of the gfx950 GEMM engine: every macro-tile (256x256, 256x128, 128x256, 128x128) x operand type
(bf16, fp16, OCP fp8 e4m3 / e5m2, int8) against an fp64 matmul of the same (de-quantised) operands, element-wise;
the in-launch split-K join; output scales; exact int32 output; and the 8-bit operand types through the fused
AG + GEMM and GEMM + RS kernels on virtual ranks.

Scenarios follow the reference's dtype options of python/triton_dist/test/amd/test_ag_gemm_intra_node.py:122-136
(float8_e4m3fn / float8_e5m2 / s8) and the BM / BN autotune space of kernels/amd/all_gather_gemm.py:102-124.
Parity with the reference's own fp8 / int8 outputs is unpinned: it holds no fixture for them and its Triton path cannot
run here; the oracle is the fp64 product of the operands the kernel was given.
"""
import pytest
import torch

from triton_distributed_gh_amd.ops import gemm as G
from triton_distributed_gh_amd.ops.ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context
from triton_distributed_gh_amd.ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node
from triton_distributed_gh_amd.utils.testing import VirtualCluster

pytestmark = pytest.mark.gpu

SHAPES = [(256, 256), (256, 128), (128, 256), (128, 128)]
DTYPES = [torch.bfloat16, torch.float16, torch.float8_e4m3fn, torch.float8_e5m2, torch.int8]


def _make(M, N, K, dt):
    if dt == torch.int8:
        return (torch.randint(-127, 128, (M, K), device="cuda", dtype=torch.int8),
                torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8))
    return ((torch.rand(M, K, device="cuda") * 2 - 1).to(dt), (torch.rand(N, K, device="cuda") * 2 - 1).to(dt))


def _close(out, ref, K, what):
    """Every element: rounding of a 16-bit result plus fp32 accumulation noise over K products."""
    tol = 2.0 ** -10 if out.dtype == torch.float16 else 2.0 ** -7
    err = (out.double() - ref).abs()
    lim = tol * ref.abs() + 1e-3 * K ** 0.5
    bad = err > lim
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements off, worst {float((err - lim).max()):.4g}"


@pytest.mark.parametrize("dt", DTYPES, ids=lambda d: str(d).split(".")[-1])
@pytest.mark.parametrize("bm,bn", SHAPES)
@pytest.mark.parametrize("M,N,K", [(300, 264, 384), (1000, 520, 1024), (130, 136, 256)])
def test_tiles_match_fp64(dt, bm, bn, M, N, K):
    a, b = _make(M, N, K, dt)
    ref = a.double() @ b.double().t()
    if dt == torch.int8:
        out = G.gemm_nt_tiles(a, b, out_dtype=torch.int32, bm=bm, bn=bn)
        assert torch.equal(out.double(), ref), "int32 accumulators must be exact"
    else:
        out = G.gemm_nt_tiles(a, b, bm=bm, bn=bn)
        assert out.dtype == (torch.bfloat16 if a.element_size() == 1 else dt)
        _close(out, ref, K, f"{dt} {bm}x{bn}")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float8_e4m3fn, torch.int8], ids=lambda d: str(d).split(".")[-1])
@pytest.mark.parametrize("bm,bn", SHAPES)
@pytest.mark.parametrize("split", [2, 3, 4])
def test_split_k_join(dt, bm, bn, split):
    """Wait-free join of the K ranges inside the launch; run twice - the second launch needs re-armed counters."""
    M, N, K = 300, 520, 4096
    a, b = _make(M, N, K, dt)
    ref = a.double() @ b.double().t()
    for _ in range(2):
        if dt == torch.int8:
            out = G.gemm_nt_tiles(a, b, out_dtype=torch.int32, bm=bm, bn=bn, split_k=split)
            assert torch.equal(out.double(), ref)
        else:
            _close(G.gemm_nt_tiles(a, b, bm=bm, bn=bn, split_k=split), ref, K, f"{dt} {bm}x{bn} split {split}")


def test_last_round_split_matches_fp64():
    """More tiles than CUs and a partial last round: whole rounds unsplit + split tail, every element."""
    M, N, K = 2560, 8192, 2048            # 320 tiles of 256x256
    a, b = _make(M, N, K, torch.bfloat16)
    ref = a.double() @ b.double().t()
    for _ in range(2):
        out = G.gemm_nt_tiles(a, b, bm=256, bn=256, split_k=4)
        _close(out, ref, K, "last-round split")
    _close(G.gemm_nt(a, b), ref, K, "gemm_nt through the plan")


@pytest.mark.parametrize("dt", [torch.float8_e4m3fn, torch.int8, torch.float16], ids=lambda d: str(d).split(".")[-1])
@pytest.mark.parametrize("bm,bn", SHAPES)
def test_output_scales(dt, bm, bn):
    """Per-row x per-column fp32 scales in the epilogue = de-quantisation of per-token / per-channel quantised operands."""
    M, N, K = 257, 520, 640
    a, b = _make(M, N, K, dt)
    sa = (torch.rand(M, device="cuda") + 0.5) * (1e-2 if dt == torch.int8 else 1.0)
    sb = (torch.rand(N, device="cuda") + 0.5) * (1e-2 if dt == torch.int8 else 1.0)
    ref = (a.double() @ b.double().t()) * sa.double()[:, None] * sb.double()[None, :]
    out = G.gemm_nt_tiles(a, b, scale_a=sa, scale_b=sb, bm=bm, bn=bn)
    _close(out, ref, K, f"scaled {dt} {bm}x{bn}")
    only_a = G.gemm_nt_tiles(a, b, scale_a=sa, bm=bm, bn=bn)
    _close(only_a, (a.double() @ b.double().t()) * sa.double()[:, None], K, f"row-scaled {dt} {bm}x{bn}")


def test_quantised_linear_tracks_the_bf16_one():
    """End to end: per-token / per-channel fp8 quantisation + scaled fp8 GEMM vs the bf16 GEMM of the unquantised operands."""
    M, N, K = 512, 768, 2048
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    sx = x.abs().amax(dim=1) / 448.0
    sw = w.abs().amax(dim=1) / 448.0
    xq = (x / sx[:, None]).to(torch.float8_e4m3fn)
    wq = (w / sw[:, None]).to(torch.float8_e4m3fn)
    out = G.gemm_nt_tiles(xq, wq, scale_a=sx, scale_b=sw).float()
    ref = x @ w.t()
    rel = (out - ref).norm() / ref.norm()
    assert rel < 0.06, f"fp8 linear is {float(rel):.3f} away from fp32 in relative Frobenius norm"


def test_cost_model_picks_a_plan_and_few_tiles_gemm_is_native(monkeypatch):
    """gemm_nt no longer hands the few-tiles band to the library: the plan picks a 128-wide tile or a split."""
    import ctypes

    from triton_distributed_gh_amd import _lib

    lib = _lib.hip()
    picks = {}
    for M, N, K in [(2048, 1024, 8192), (2048, 4096, 4096), (4096, 8192, 8192), (4096, 5120, 25600)]:
        bm, bn, sk, pb, nc, nf = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong(), ctypes.c_int(), ctypes.c_int()
        _lib.check(lib.tdist_gemm_nt_ex_plan(M, N, K, 0, 0, 0, 0, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sk),
                                             ctypes.byref(pb), ctypes.byref(nc), ctypes.byref(nf)), "plan")
        picks[(M, N, K)] = (bm.value, bn.value, sk.value, nf.value)
    assert picks[(4096, 8192, 8192)][:3] == (256, 256, 1)
    # 320 tiles on 256 CUs: one whole round unsplit, the 64 tiles of the partial round cut along K
    assert picks[(4096, 5120, 25600)][2] > 1 and picks[(4096, 5120, 25600)][3] == 256
    assert picks[(2048, 1024, 8192)][2] > 1                      # 32 big tiles for 256 CUs: K is split
    assert picks[(2048, 4096, 4096)][:2] != (256, 256)           # 128 big tiles: a 128-wide tile fills the chip
    called = []
    monkeypatch.setattr(torch, "matmul", lambda *a, **k: called.append(1))
    a, b = _make(2048, 1024, 8192, torch.bfloat16)
    out = G.gemm_nt(a, b)
    assert not called, "a supported few-tiles GEMM went to the library"
    _close(out, a.double() @ b.double().t(), 8192, "few-tiles gemm_nt")


# ---------------------------------------------------------------------------------------------------------------
# 8-bit operands through the fused tensor-parallel kernels (virtual ranks on one GPU)

def _ctas(w):
    return max(8, 200 // w)


@pytest.mark.parametrize("dt", [torch.float8_e4m3fn, torch.float8_e5m2, torch.int8], ids=lambda d: str(d).split(".")[-1])
@pytest.mark.parametrize("w,M,N,K,tile", [(2, 1024, 1024, 512, ""), (4, 2048, 2048, 1024, ""), (4, 512, 1024, 512, "128x256"),
                                          (4, 1024, 1024, 2048, "256x128")])
def test_ag_gemm_8bit_virtual_ranks(dt, w, M, N, K, tile, monkeypatch):
    if tile:
        monkeypatch.setenv("TDIST_AG_TILE", tile)
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ctxs = vc.collective(lambda r, env: create_ag_gemm_intra_node_context(
            M, N // w, K, dt, torch.bfloat16, r, w, num_ctas=_ctas(w), env=env))
        for it in range(2):
            ab = [_make(M // w, N // w, K, dt) for _ in range(w)]
            sa = torch.rand(M, device="cuda") + 0.5 if it else None
            sb = [torch.rand(N // w, device="cuda") + 0.5 for _ in range(w)] if it else [None] * w
            if dt == torch.int8 and it:
                sa, sb = sa * 1e-2, [s * 1e-2 for s in sb]
            torch.cuda.synchronize()
            outs = vc.run(lambda r, env: ag_gemm_intra_node(ab[r][0], ab[r][1], False, ctxs[r], scale_a=sa, scale_b=sb[r]))
            a_full = torch.cat([x[0] for x in ab]).double()
            for r in range(w):
                ctxs[r].check_error()
                ref = a_full @ ab[r][1].double().t()
                if it:
                    ref = ref * sa.double()[:, None] * sb[r].double()[None, :]
                assert outs[r].dtype == torch.bfloat16
                if dt == torch.int8 and not it:
                    # unscaled int8: the bf16 output rounds the exact int32 sums
                    assert bool(((outs[r].double() - ref).abs() <= 2.0 ** -7 * ref.abs() + 1.0).all())
                else:
                    _close(outs[r], ref, K, f"ag_gemm {dt} rank {r} iteration {it}")
    finally:
        vc.close()


@pytest.mark.parametrize("dt", [torch.float8_e4m3fn, torch.int8], ids=lambda d: str(d).split(".")[-1])
@pytest.mark.parametrize("w,M,N,K", [(2, 1024, 1024, 1024), (4, 2048, 1024, 2048)])
def test_gemm_rs_8bit_virtual_ranks(dt, w, M, N, K):
    vc = VirtualCluster(w, heap_bytes=1 << 27)
    try:
        ctxs = vc.collective(lambda r, env: create_gemm_rs_intra_node_context(
            M, N, torch.bfloat16, r, w, num_ctas=_ctas(w), env=env))
        for it in range(2):
            ab = [_make(M, N, K // w, dt) for _ in range(w)]
            scale = 1e-3 if dt == torch.int8 else 1.0
            sa = (torch.rand(M, device="cuda") + 0.5) * scale
            sb = torch.rand(N, device="cuda") + 0.5
            torch.cuda.synchronize()
            outs = vc.run(lambda r, env: gemm_rs_intra_node(ab[r][0], ab[r][1], ctxs[r], scale_a=sa, scale_b=sb))
            parts = [((x.double() @ y.double().t()) * sa.double()[:, None] * sb.double()[None, :]) for x, y in ab]
            total = sum(parts)
            mag = sum(p.abs() for p in parts)
            Mr = M // w
            for r in range(w):
                ctxs[r].check_error()
                ref = total[r * Mr:(r + 1) * Mr]
                err = (outs[r].double() - ref).abs()
                # every partial and the sum are rounded to bf16 once
                lim = 2.0 ** -7 * (mag[r * Mr:(r + 1) * Mr] + ref.abs()) + 1e-3 * (K // w) ** 0.5
                assert bool((err <= lim).all()), f"gemm_rs {dt} rank {r}: worst {float((err - lim).max()):.4g}"
    finally:
        vc.close()


def test_transposed_weight_is_cached():
    """transe_b=True: one K-contiguous copy per weight, not one per call."""
    from triton_distributed_gh_amd.ops.ag_gemm import k_contiguous_weight

    wt = torch.randn(512, 256, device="cuda").bfloat16()          # [K, N]
    first = k_contiguous_weight(wt)
    assert first.shape == (256, 512) and first.stride(1) == 1
    assert k_contiguous_weight(wt) is first
    wt.mul_(2)                                                     # in-place update: a new copy
    second = k_contiguous_weight(wt)
    assert second is not first and torch.equal(second, wt.t())
    assert k_contiguous_weight(first.t()).data_ptr() == first.data_ptr()   # a transposed view costs nothing
