"""gfx950 GEMM kernel vs an fp32 PyTorch reference (scenario of the reference's
python/triton_dist/test/amd/test_matmul_amd.py)."""
import pytest
import torch

from triton_distributed_gh_amd.ops.gemm import gemm_nt
from triton_distributed_gh_amd.utils.testing import assert_close_elementwise

pytestmark = pytest.mark.gpu

SHAPES = [(256, 256, 128), (512, 768, 256), (300, 520, 192), (1000, 1032, 1024), (64, 4096, 512), (4096, 8192, 1024)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_nt_matches_fp32_reference(M, N, K, dtype):
    torch.manual_seed(M * 7 + N * 3 + K)
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dtype)
    b = (torch.rand(N, K, device="cuda") * 2 - 1).to(dtype)
    c = gemm_nt(a, b, force_native=True)
    ref = a.float() @ b.float().t()
    rel = (c.float() - ref).abs().max().item() / ref.abs().max().item()
    assert rel < 1e-2
    assert_close_elementwise(c, ref, what=f"gemm_nt {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", SHAPES + [(4096, 4096, 512), (1000, 8200, 256)])
@pytest.mark.parametrize("knobs", [0, 3])
def test_persistent_gemm_matches_fp32_reference(M, N, K, knobs):
    """Variant 3: persistent workgroups, permuted accumulator columns, stores straight from registers
    (more than one tile per workgroup at the larger shapes, ragged edges, untouched padding)."""
    torch.manual_seed(M + N * 5 + K)
    for dtype in (torch.bfloat16, torch.float16):
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dtype)
        b = (torch.rand(N, K, device="cuda") * 2 - 1).to(dtype)
        out = torch.full((M, N + 8), 7.0, device="cuda", dtype=dtype)
        gemm_nt(a, b, out=out[:, :N], force_native=True, variant=3 | (knobs << 16))
        ref = a.float() @ b.float().t()
        rel = (out[:, :N].float() - ref).abs().max().item() / ref.abs().max().item()
        assert rel < 1e-2
        assert_close_elementwise(out[:, :N], ref, what=f"gemm_nt variant {M}x{N}x{K}")
        assert torch.all(out[:, N:] == 7.0), "wrote past the last column"


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 520, 192), (1000, 8200, 256), (4096, 4096, 512)])
def test_four_wave_gemm_matches_fp32_reference(M, N, K):
    """Variant 4: four 128x128 waves, accumulators in the accumulation registers, slab ring of four,
    stores straight from registers (ragged edges, several tiles per workgroup, untouched padding)."""
    torch.manual_seed(M + N * 3 + K)
    for dtype in (torch.bfloat16, torch.float16):
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dtype)
        b = (torch.rand(N, K, device="cuda") * 2 - 1).to(dtype)
        out = torch.full((M, N + 8), 7.0, device="cuda", dtype=dtype)
        gemm_nt(a, b, out=out[:, :N], force_native=True, variant=4)
        ref = a.float() @ b.float().t()
        rel = (out[:, :N].float() - ref).abs().max().item() / ref.abs().max().item()
        assert rel < 1e-2
        assert_close_elementwise(out[:, :N], ref, what=f"gemm_nt variant {M}x{N}x{K}")
        assert torch.all(out[:, N:] == 7.0), "wrote past the last column"


def test_persistent_gemm_column_permutation_is_undone():
    # A = I: the output must be exactly B^T, so any column left in the permuted order shows up.
    K = 256
    a = torch.eye(K, device="cuda", dtype=torch.bfloat16)
    b = torch.arange(512 * K, device="cuda", dtype=torch.float32).reshape(512, K).remainder(251).bfloat16()
    c = gemm_nt(a, b, force_native=True, variant=3)
    assert torch.equal(c, b.t().contiguous())


def test_gemm_nt_asymmetric_identity():
    # A = I with an asymmetric B catches a transposed output or operand mapping.
    K = 256
    a = torch.eye(K, device="cuda", dtype=torch.bfloat16)
    b = torch.arange(512 * K, device="cuda", dtype=torch.float32).reshape(512, K).remainder(251).bfloat16()
    c = gemm_nt(a, b, force_native=True)
    assert torch.equal(c, b.t().contiguous())


def test_gemm_nt_strided_views():
    big_a = (torch.rand(512, 1024, device="cuda") - 0.5).bfloat16()
    big_b = (torch.rand(768, 1024, device="cuda") - 0.5).bfloat16()
    a, b = big_a[:, :512], big_b[:, 256:768]
    out = torch.zeros(512, 1024, device="cuda", dtype=torch.bfloat16)
    gemm_nt(a, b, out=out[:, :768], force_native=True)
    ref = a.float() @ b.float().t()
    assert (out[:, :768].float() - ref).abs().max().item() < 0.5
    assert out[:, 768:].abs().max().item() == 0


def test_gemm_fallback_for_unsupported_shape():
    a = torch.rand(17, 100, device="cuda").bfloat16()
    b = torch.rand(24, 100, device="cuda").bfloat16()
    c = gemm_nt(a, b)
    assert torch.allclose(c.float(), a.float() @ b.float().t(), atol=0.5)
    with pytest.raises(ValueError):
        gemm_nt(a, b, force_native=True)


@pytest.mark.parametrize("M", [1, 5, 16])
@pytest.mark.parametrize("N,K", [(4096, 4096), (12800, 2048), (1000, 192), (32768, 1024), (40, 128)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_decode_gemm_matches_fp32_reference(M, N, K, dtype):
    """M <= 16 takes the weight-streaming decode kernel: every column-block size (16 / 32 / 64), a K
    that is not a multiple of the 8-step unroll, partial column blocks, strided activations."""
    torch.manual_seed(M + N + K)
    xa = (torch.rand(M, K + 64, device="cuda") * 2 - 1).to(dtype)
    a = xa[:, :K]                                       # row stride K + 64
    b = (torch.rand(N, K, device="cuda") * 2 - 1).to(dtype)
    out = torch.full((M, N + 8), 7.0, device="cuda", dtype=dtype)
    gemm_nt(a, b, out=out[:, :N], force_native=True)
    ref = a.float() @ b.float().t()
    rel = (out[:, :N].float() - ref).abs().max().item() / ref.abs().max().item()
    assert rel < 1e-2
    assert torch.all(out[:, N:] == 7.0), "wrote past the last column"
    # same numbers as the tile kernel within rounding (debug bit 1 of the variant forces it)
    c2 = gemm_nt(a, b, force_native=True, variant=2 | (2 << 8))
    assert (c2.float() - out[:, :N].float()).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("silu", [0, 1])
@pytest.mark.parametrize("nw", [0, 1, 2, 4, 8])
@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (1, 1000, 2080), (1, 12, 160), (2, 520, 1056), (1, 6, 128)])
def test_one_row_gemm_on_the_vector_alus(M, N, K, nw, silu):
    """gemm_nt_dot_kernel (what gemm_nt / gemm_nt_norm / gemm_nt_silu_mul run at one row) with every K split, ragged
    N and K that is not a multiple of the 64-element step, against fp32 PyTorch."""
    import ctypes
    from triton_distributed_gh_amd import _lib

    torch.manual_seed(N + K + nw)
    bf = torch.bfloat16
    rows = 2 * N if silu else N
    x = (torch.rand(M, K, device="cuda") * 2 - 1).to(bf)
    w = ((torch.rand(rows, K, device="cuda") * 2 - 1) * 0.1).to(bf)
    out = torch.zeros(M, N, device="cuda", dtype=bf)
    _lib.check(_lib.hip().tdist_gemm_nt_dot(_lib.ptr(x), _lib.ptr(w), _lib.ptr(out), M, N, K, ctypes.c_longlong(K),
                                            ctypes.c_longlong(K), ctypes.c_longlong(N), silu, nw, _lib.stream_handle()),
               "tdist_gemm_nt_dot")
    ref = x.float() @ w.float().t()
    if silu:
        ref = torch.nn.functional.silu(ref[:, :N].to(bf).float()) * ref[:, N:].to(bf).float()
    torch.testing.assert_close(out.float(), ref, atol=2e-2 * max(ref.abs().max().item(), 1e-3), rtol=2e-2)


@pytest.mark.parametrize("M", [1, 2, 9, 16, 40])
@pytest.mark.parametrize("N,K", [(4096, 4096), (6144, 2048), (12800, 8192), (776, 192), (50251, 1024), (100104, 3072)])
def test_linear_with_bias(M, N, K):
    """``gemm_nt(..., bias=)``: the bias rides in the epilogue of the decode kernels (one row on the vector ALUs, 2 .. 16
    rows on MFMAs, the eight-slice shape of long-K layers included), in the epilogue of the 128 x 256 tiles that take
    17 .. 128 rows of an LM head with more than 100k columns, and is a separate add elsewhere (odd N: the library's own
    epilogue); the same numbers as fp32 PyTorch either way."""
    from triton_distributed_gh_amd.ops.gemm import gemm_nt

    torch.manual_seed(M + N + K)
    bf = torch.bfloat16
    x = (torch.rand(M, K, device="cuda") * 2 - 1).to(bf)
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.1).to(bf)
    bias = (torch.rand(N, device="cuda") * 4 - 2).to(bf)
    out = gemm_nt(x, w, bias=bias)
    ref = x.float() @ w.float().t() + bias.float()
    torch.testing.assert_close(out.float(), ref, atol=2e-2 * max(ref.abs().max().item(), 1e-3), rtol=2e-2)
    assert_close_elementwise(out, ref, what=f"linear + bias {M}x{N}x{K}")


@pytest.mark.parametrize("M", [1, 7, 16])
@pytest.mark.parametrize("I,K", [(1024, 512), (12288, 4096), (1000, 192)])
def test_decode_gemm_with_silu_mul_epilogue(M, I, K):
    """gate_up projection + silu(gate) * up in one launch vs the two separate kernels and fp32."""
    from triton_distributed_gh_amd.ops.elementwise import silu_mul
    from triton_distributed_gh_amd.ops.gemm import gemm_nt_silu_mul

    torch.manual_seed(M + I + K)
    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand(2 * I, K, device="cuda") * 2 - 1) * 0.1).bfloat16()
    got = gemm_nt_silu_mul(a, w)
    two = silu_mul(gemm_nt(a, w, force_native=True))
    gu = a.float() @ w.float().t()
    ref = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]
    assert got.shape == (M, I)
    assert (got.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item() + 1e-3
    assert (got.float() - two.float()).abs().max().item() <= 1e-2 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("M", [1, 3, 5])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 4096), (1000, 192), (16384, 1024), (40, 128)])
@pytest.mark.parametrize("with_residual", [False, True])
def test_decode_gemm_with_fused_rms_norm(M, N, K, with_residual, monkeypatch):
    """RMSNorm (+ residual add) folded into the decode GEMM vs the two kernels and vs fp32: every column-block
    size, partial column blocks, a K below one batch of loads per wave, strided activations."""
    from triton_distributed_gh_amd.ops import gemm as gemm_mod
    from triton_distributed_gh_amd.ops.elementwise import rms_norm
    from triton_distributed_gh_amd.ops.gemm import gemm_nt_norm, norm_gemm_supported

    monkeypatch.setattr(gemm_mod, "NORM_GEMM_MAX_M", 16)     # (the dispatch fuses up to two rows; the kernel takes 16)

    torch.manual_seed(M + N + K)
    xa = ((torch.rand(M, K + 64, device="cuda") * 2 - 1) * 3).bfloat16()
    x = xa[:, :K]
    res = ((torch.rand(M, K, device="cuda") * 2 - 1) * 3).bfloat16() if with_residual else None
    g = (torch.rand(K, device="cuda") + 0.5).bfloat16()
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.1).bfloat16()
    assert norm_gemm_supported(x, w, g, res)
    res_before = res.clone() if with_residual else None
    out, hidden = gemm_nt_norm(x, g, 1e-5, w, residual=res)
    if with_residual:
        y2, h2 = rms_norm(x.contiguous(), g, 1e-5, residual=res)
        assert torch.equal(hidden, h2), "x + residual must be bit-identical to the norm kernel's"
        assert torch.equal(res, res_before)
    else:
        y2 = rms_norm(x.contiguous(), g, 1e-5)
        assert hidden is x
    two = gemm_nt(y2, w, force_native=True)
    hf = (x.float() + res.float()).bfloat16().float() if with_residual else x.float()
    ref = (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + 1e-5) * g.float()) @ w.float().t()
    scale = ref.abs().max().item()
    assert (out.float() - ref).abs().max().item() <= 2e-2 * scale + 1e-3
    assert (out.float() - two.float()).abs().max().item() <= 2e-2 * scale + 1e-3


@pytest.mark.parametrize("M", [1, 4])
def test_decode_gemm_with_fused_rms_norm_and_silu(M, monkeypatch):
    from triton_distributed_gh_amd.ops import gemm as gemm_mod
    from triton_distributed_gh_amd.ops.gemm import gemm_nt_norm

    monkeypatch.setattr(gemm_mod, "NORM_GEMM_MAX_M", 16)

    torch.manual_seed(M)
    K, I = 4096, 12288
    x = ((torch.rand(M, K, device="cuda") * 2 - 1) * 2).bfloat16()
    res = ((torch.rand(M, K, device="cuda") * 2 - 1) * 2).bfloat16()
    g = (torch.rand(K, device="cuda") + 0.5).bfloat16()
    w = ((torch.rand(2 * I, K, device="cuda") * 2 - 1) * 0.05).bfloat16()
    out, hidden = gemm_nt_norm(x, g, 1e-6, w, residual=res, silu=True)
    hf = (x.float() + res.float()).bfloat16().float()
    assert torch.equal(hidden.float(), hf)
    gu = (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + 1e-6) * g.float()) @ w.float().t()
    ref = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]
    assert out.shape == (M, I)
    assert (out.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item() + 1e-3


def test_fused_rms_norm_gemm_falls_back_outside_its_shapes():
    """More than 16 rows (or rows x K beyond the LDS buffer): the norm kernel followed by the GEMM, same contract."""
    from triton_distributed_gh_amd.ops.gemm import gemm_nt_norm, norm_gemm_supported

    torch.manual_seed(0)
    x = (torch.rand(16, 4096, device="cuda") * 2 - 1).bfloat16()
    g = torch.ones(4096, device="cuda", dtype=torch.bfloat16)
    w = ((torch.rand(512, 4096, device="cuda") * 2 - 1) * 0.1).bfloat16()
    assert not norm_gemm_supported(x, w, g, None)           # 16 x 8 KiB of activations do not fit
    out, hidden = gemm_nt_norm(x, g, 1e-6, w)
    xf = x.float()
    ref = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)) @ w.float().t()
    assert hidden is x and (out.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


def test_measured_gemm_times_sit_between_the_speed_of_light_and_the_model():
    """The perf model's MI355X tables against this GPU: a measured GEMM can never beat the speed-of-light
    estimate, and stays within 2x of what the model expects of the tile engine / the streaming kernel."""
    from triton_distributed_gh_amd.utils import perf_model as pm

    def timed_ms(fn, iters=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    for M, N, K in [(4096, 8192, 8192), (1, 32768, 4096)]:
        a = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
        b = (torch.rand(N, K, device="cuda") - 0.5).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t = timed_ms(lambda: gemm_nt(a, b, out=out, force_native=True))
        sol, exp = pm.estimate_gemm_sol_time_ms(M, N, K), pm.estimate_gemm_time_ms(M, N, K)
        assert sol <= t <= 2.0 * exp, f"{M}x{N}x{K}: measured {t:.4f} ms, speed of light {sol:.4f}, model {exp:.4f}"


@pytest.mark.parametrize("M,N,K,split", [(2048, 2048, 4096, 0), (1300, 1024, 2048, 0), (2048, 520, 1024, 2),
                                         (4000, 2048, 8192, 4), (1536, 1536, 640, 3)])
def test_splitk_gemm_matches_fp32_reference(M, N, K, split):
    """Few 256x256 tiles: K cut into ranges, the last item of a tile sums the parked fp32 accumulators.
    Element-wise against the fp32 reference, twice in a row (the kernel re-arms its own counters), and
    through gemm_nt's routing for the shapes it sends there."""
    from triton_distributed_gh_amd.ops.gemm import gemm_nt_splitk

    torch.manual_seed(M + N + K)
    a = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
    b = (torch.rand(N, K, device="cuda") - 0.5).bfloat16()
    ref = a.float() @ b.float().t()
    for _ in range(2):
        c = gemm_nt_splitk(a, b, split_k=split)
        torch.cuda.synchronize()
        torch.testing.assert_close(c.float(), ref, atol=2e-2 * ref.abs().max().item() / 4, rtol=2e-2)
    for c2 in (gemm_nt(a, b), gemm_nt(a, b, force_native=True)):      # default routing and the 256x256 tiles
        torch.testing.assert_close(c2.float(), ref, atol=2e-2 * ref.abs().max().item() / 4, rtol=2e-2)


def test_splitk_gemm_replays_from_a_graph():
    from triton_distributed_gh_amd.ops.gemm import gemm_nt_splitk

    a = (torch.rand(2048, 4096, device="cuda") - 0.5).bfloat16()
    b = (torch.rand(1024, 4096, device="cuda") - 0.5).bfloat16()
    gemm_nt_splitk(a, b)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s):
        c = gemm_nt_splitk(a, b)
    for _ in range(3):
        a.copy_((torch.rand(2048, 4096, device="cuda") - 0.5).bfloat16())
        g.replay()
        torch.cuda.synchronize()
        ref = a.float() @ b.float().t()
        torch.testing.assert_close(c.float(), ref, atol=2e-2 * ref.abs().max().item() / 4, rtol=2e-2)


def test_table_of_measured_forms_dispatches_correctly():
    """Every cell of the shipped table (tuned/gemm_mi355x.json) with a weight of at most 128 MB: ``gemm_nt`` takes the
    table's form for it and the result matches fp32 PyTorch; the look-up falls back to the rules for shapes and row
    counts the table does not hold."""
    from triton_distributed_gh_amd.ops import gemm as G

    table = G._gemm_table()
    assert table, "the table of measured forms is part of the package"
    bf = torch.bfloat16
    checked = 0
    for (N, K), cells in sorted(table.items()):
        if N * K * 2 > (128 << 20):
            continue
        w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(bf)
        for M, form in sorted(cells.items()):
            assert G._tuned_form(M, N, K) == form
            x = (torch.rand(M, K, device="cuda") * 2 - 1).to(bf)
            out = torch.full((M, N), float("nan"), device="cuda", dtype=bf)
            assert G.run_gemm_form(form, x, w, out), (N, K, M, form)
            ref = x.float() @ w.float().t()
            torch.testing.assert_close(out.float(), ref, atol=2e-2 * ref.abs().max().item(), rtol=2e-2)
            got = G.gemm_nt(x, w)               # (split-K forms add their partial sums in arrival order: not bit-equal)
            torch.testing.assert_close(got.float(), ref, atol=2e-2 * ref.abs().max().item(), rtol=2e-2)
            checked += 1
    assert checked >= 20
    assert G._tuned_form(7, 123456, 4096) is None and G._tuned_form(5000, 6144, 4096) is None
