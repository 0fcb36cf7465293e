"""Single-GPU 16-bit GEMM on the gfx950 MFMA tile engine:  C = A @ B^T.

``gemm_nt(a, b)`` takes ``a [M, K]`` and ``b [N, K]`` (both K-contiguous - the layout every fused
op in this package uses for weights) and returns ``[M, N]``.  M <= 16 runs the decode kernel
(weights streamed once through 16x16x32 MFMAs, K split inside the workgroup), everything else the
256x256 tile kernel.  Shapes the hand-written kernel does
not cover (K % 64 != 0, K < 128, N % 8 != 0, other dtypes) and CPU tensors go to ``torch.matmul``.

Role in the reference: Triton matmul of python/triton_dist/test/amd/test_matmul_amd.py and the
GEMM halves of kernels/amd/all_gather_gemm.py / gemm_reduce_scatter.py.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from .. import _lib
from ..autotuner import tuning

_DT = {torch.bfloat16: 0, torch.float16: 1}
# tile kernel: 1 = four phases per K-step, 2 = two phases per K-step (one tile per workgroup, LDS-staged
# epilogue), 3 = two phases, persistent workgroups, register-direct epilogue (bits 16-17: schedule knobs,
# see mainloop_256_2ph), 4 = four 128x128 waves.  Default: 3 with both knobs - 0.5-0.8 % ahead of 2 in
# every interleaved run at 4096x8192x8192 (profiles/gemm_persistent_r1.md).
GEMM_VARIANT = int(os.environ.get("TDIST_GEMM_VARIANT", str(3 | (3 << 16))))
# decode layers (M <= 16, K >= 4096) with at least this many MiB of weights go to the library (0 = never)
DECODE_LIB_BYTES = int(os.environ.get("TDIST_DECODE_GEMM_LIB_MB", "0")) << 20
# the band between the decode kernel and chip-filling 256x256 tiles: "native" (mid-M kernel) or "lib"
MID_GEMM = os.environ.get("TDIST_MID_GEMM", "native")
MID_MAX_M = int(os.environ.get("TDIST_MID_MAX_M", "1024"))
# 0: the whole band up to MID_MAX_M rows on the mid-M kernel (A/B comparisons); default: above 128 rows the tile family
MID_BAND_TILES = os.environ.get("TDIST_MID_BAND_TILES", "1") != "0"
# decode shapes (M <= 16) with long K / mid-sized weights on the mid-M kernel's split-K: from this many rows on
DECODE_SPLITK = os.environ.get("TDIST_DECODE_SPLITK", "0") != "0"
DECODE_SPLITK_MIN_M = 1 if DECODE_SPLITK else int(os.environ.get("TDIST_DECODE_SPLITK_MIN_M", "9"))


def gemm_supported(a: torch.Tensor, b: torch.Tensor) -> bool:
    if not (a.is_cuda and b.is_cuda) or a.dtype not in _DT or a.dtype != b.dtype:
        return False
    if a.dim() != 2 or b.dim() != 2 or a.stride(1) != 1 or b.stride(1) != 1:
        return False
    M, K = a.shape
    N = b.shape[0]
    return (K >= 128 and K % 64 == 0 and N % 8 == 0 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and M > 0 and N > 0)


# the 128-byte-run decode kernels (the ones with a bias epilogue) are on unless switched off for A/B measurements
_DECODE_BIAS_KERNELS = os.environ.get("TDIST_SKINNY128", "") != "0" and os.environ.get("TDIST_DECODE_DOT", "") != "0"


# ---- measured forms per shape ---------------------------------------------------------------------------------------
# `scripts/tune_gemm_table.py` times every kernel this module has for a list of (N, K) weight shapes at one row count
# per bucket (cold weights) and writes the winners that beat the rules below by more than a few per cent to a JSON
# file: {"N,K": {"<rows of the bucket's upper end>": "<form>"}}.  Forms: "dot:<K slices>" (one row), "s128:<WN><NW>"
# (2 .. 16 rows), "mid" / "mid:<wn>:<splits>", "tile:<bm>x<bn>" / "tile:<bm>x<bn>/<K split>", "tiles" (the tile family's
# own plan), "native256".  The table shipped with the package covers the projection shapes of fifteen models (Qwen3,
# Qwen2.5, Llama-3.x, Seed-OSS-36B, Gemma-2-9B, Mistral-Nemo-12B - FAMILIES of the script - whole and, for the large
# ones, as TP = 8 shards); TDIST_GEMM_TABLE names another file, "0" switches the look-up off.
# Role in the reference: the tuned-config caches behind triton.autotune / the contextual autotuner (autotuner.py).
GEMM_BUCKETS = (1, 4, 8, 16, 32, 64, 128, 192, 256, 384, 512, 768, 1024, 2048)
_TABLE = None


def _gemm_table() -> dict:
    global _TABLE
    if _TABLE is None:
        _TABLE = {}
        path = os.environ.get("TDIST_GEMM_TABLE", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tuned",
                                                                "gemm_mi355x.json"))
        if path != "0" and os.path.exists(path):
            import json

            with open(path) as f:
                raw = json.load(f)
            _TABLE = {tuple(int(v) for v in k.split(",")): {int(m): form for m, form in cells.items()}
                      for k, cells in raw.items() if "," in k}
    return _TABLE


def _tuned_form(M: int, N: int, K: int) -> Optional[str]:
    cells = _gemm_table().get((N, K))
    if not cells:
        return None
    for top in GEMM_BUCKETS:
        if M <= top:
            return cells.get(top)
    return None


def run_gemm_form(form: str, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, group_m: int = 8) -> bool:
    """Run ``a @ b^T`` into ``out`` on the kernel a table entry names; False if the form does not apply to this call
    (wrong row count for the kernel, unsupported operands) - the caller then goes by its rules."""
    M, K = a.shape
    N = b.shape[0]
    kind, _, arg = form.partition(":")
    lib = _lib.hip()
    bf16 = a.dtype == torch.bfloat16
    if kind == "dot":
        if not (bf16 and M <= 2 and K % 32 == 0):
            return False
        _lib.check(lib.tdist_gemm_nt_dot(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), M, N, K, ctypes.c_longlong(a.stride(0)),
                                         ctypes.c_longlong(b.stride(0)), ctypes.c_longlong(out.stride(0)), 0, int(arg),
                                         _lib.stream_handle()), "tdist_gemm_nt_dot")
        return True
    if kind == "s128":
        if not (bf16 and 2 <= M <= 16):
            return False
        _lib.check(lib.tdist_gemm_nt_skinny128(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), M, N, K, ctypes.c_longlong(a.stride(0)),
                                               ctypes.c_longlong(b.stride(0)), ctypes.c_longlong(out.stride(0)), int(arg[0]),
                                               int(arg[1]), _lib.stream_handle()), "tdist_gemm_nt_skinny128")
        return True
    if kind == "mid":
        if M > MID_MAX_M:
            return False
        wn, _, sp = arg.partition(":")
        gemm_nt_mid(a, b, out, wn=int(wn or 0), splits=int(sp or 0))
        return True
    if kind in ("tile", "tiles"):
        if not gemm_tiles_supported(a, b):
            return False
        if kind == "tiles":
            gemm_nt_tiles(a, b, out, group_m=group_m)
            return True
        shape, _, sk = arg.partition("/")
        bm, _, bn = shape.partition("x")
        gemm_nt_tiles(a, b, out, bm=int(bm), bn=int(bn), split_k=int(sk or 0), group_m=group_m)
        return True
    if kind == "native256":
        _lib.check(lib.tdist_gemm_nt(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), M, N, K, ctypes.c_longlong(a.stride(0)),
                                     ctypes.c_longlong(b.stride(0)), ctypes.c_longlong(out.stride(0)), _DT[a.dtype], group_m,
                                     _lib.stream_handle(), GEMM_VARIANT), "tdist_gemm_nt")
        return True
    return False


def _big_head(M: int, N: int, K: int, variant, force_native: bool) -> bool:
    """17 .. 128 rows against more than 100k columns with K >= 3072: the band ``gemm_nt`` gives to 128 x 256 tiles."""
    return (16 < M <= 128 and (variant is None or variant == GEMM_VARIANT) and N >= 100_000 and K >= 3072
            and not force_native)


def _decode_takes_mid(M: int, N: int, K: int) -> bool:
    """Decode shapes (M <= 16) that run on the mid-M kernel instead of the weight-streaming ones."""
    if M > 16:
        return False
    if DECODE_SPLITK_MIN_M <= M and N < 12288 and K >= 8192:
        return True
    if DECODE_SPLITK_MIN_M > 16 or M < 2:
        return False
    if 4096 < N < 6144:
        return K >= 5120 or (M >= 9 and K >= 3072)
    # (the same with 32-column blocks between 8192 and 16384 columns: 10240 / 12288 / 15360 x 5120, 12288 x 3072 and
    # 14336 x 3584 at 16 rows 1.15 - 1.24 x; with K = 4096 the 64-column streaming shape stays ahead)
    return M >= 9 and 8192 < N < 16384 and K >= 3072 and K % 4096 != 0


def gemm_nt(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, group_m: Optional[int] = None,
            force_native: bool = False, variant: Optional[int] = None, bias: Optional[torch.Tensor] = None,
            use_table: bool = True) -> torch.Tensor:
    """``a [M, K] @ b [N, K]^T`` (+ ``bias [N]``) -> [M, N] on the gfx950 kernels; shapes they do not take run on
    ``torch.matmul``.  The bias is added in the epilogue of the weight-streaming decode kernels (bf16, M <= 16) and as
    a separate in-place add everywhere else.  ``use_table=False`` ignores the table of measured forms (the tuner's
    "what the rules pick" column)."""
    assert a.shape[1] == b.shape[1], f"K mismatch: {tuple(a.shape)} vs {tuple(b.shape)}"
    if bias is not None:
        M, K = a.shape
        N = b.shape[0]
        assert bias.shape == (N,) and bias.dtype == a.dtype, "bias must be [N] of the operands' type"
        if out is None:
            out = torch.empty((M, N), dtype=a.dtype, device=a.device)
        fused = (_DECODE_BIAS_KERNELS and M <= 16 and a.dtype == torch.bfloat16 and gemm_supported(a, b)
                 and bias.is_contiguous() and out.stride(1) == 1 and out.stride(0) % 8 == 0 and out.data_ptr() % 16 == 0
                 and not _decode_takes_mid(M, N, K)
                 and not (DECODE_LIB_BYTES > 0 and K >= 4096 and N * K * 2 >= DECODE_LIB_BYTES))
        if fused:
            _lib.check(
                _lib.hip().tdist_gemm_nt_decode_bias(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.ptr(bias), M, N, K,
                                                     ctypes.c_longlong(a.stride(0)), ctypes.c_longlong(b.stride(0)),
                                                     ctypes.c_longlong(out.stride(0)), _lib.stream_handle()),
                "tdist_gemm_nt_decode_bias")
            return out
        if not gemm_supported(a, b) and not force_native:
            return torch.addmm(bias, a, b.t(), out=out)       # (the library adds it in its own epilogue)
        if _big_head(M, N, K, variant, force_native) and gemm_tiles_supported(a, b) and bias.is_contiguous():
            return gemm_nt_tiles(a, b, out, bm=128, bn=256, group_m=group_m, bias=bias)
        out = gemm_nt(a, b, out, group_m, force_native, variant)
        return out.add_(bias)
    if group_m is None:
        group_m = tuning.get("gemm.group_m", 8)
    if variant is None:
        variant = tuning.get("gemm.variant", GEMM_VARIANT)
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    if not gemm_supported(a, b) or out.stride(1) != 1 or out.stride(0) % 8 or out.data_ptr() % 16:
        if force_native:
            raise ValueError("shape/dtype not supported by the gfx950 GEMM kernel")
        torch.matmul(a, b.t(), out=out)
        return out
    # Between the decode kernel (M <= 16, weights streamed once) and shapes that fill the chip with
    # 256x256 tiles lies a band (17 .. a few hundred rows, or few column tiles) where the big tile leaves
    # most CUs idle: the weight-streaming mid-M kernel with split-K (csrc/ops/gemm_mid.hip) takes it.
    # TDIST_MID_GEMM=lib hands the band to the library instead (A/B comparisons).
    # Up to 64 rows the mid-M kernel also beats the 256x256 tile at any width (LM heads: 42 vs 76 us at
    # 32768 x 4096, 144 vs 173 us at 151936 x 2048), up to 128 rows below ~100k columns.
    if variant == GEMM_VARIANT and not force_native and use_table:
        form = _tuned_form(M, N, K)
        if form is not None and run_gemm_form(form, a, b, out, group_m):
            return out
    # LM heads with more than 100k columns and K >= 3072 at 17 .. 128 rows: 128 x 256 tiles of the tile family stream the
    # weights once with half the padding of a 256-row tile and none of the split-K traffic (128256 x 4096 at 32 / 64 /
    # 128 rows: 183 / 188 / 200 us against 229 / 235 / 257 us mid-M and 201 / 213 / 237 us hipBLASLt; 256000 x 3072:
    # 279 / 300 / 326 against 303 / 342 / 388 and 283 / 298 / 325; scripts/dev/sweep_linear_kernels.py).
    if _big_head(M, N, K, variant, force_native) and gemm_tiles_supported(a, b):
        return gemm_nt_tiles(a, b, out, bm=128, bn=256, group_m=group_m)
    if M > 16 and variant == GEMM_VARIANT and (-(-M // 256) * -(-N // 256) < 128 or M <= 64
                                               or (M <= 128 and N < 100_000)):
        if MID_GEMM == "lib" and not force_native:
            torch.matmul(a, b.t(), out=out)
            return out
        # Above 128 rows the mid-M kernel re-streams the weights once per 128-row pass and the tile family is ahead
        # except on narrow outputs at up to 256 rows (cold weights, scripts/dev/sweep_linear_kernels.py: 6144 x 4096 at
        # 384 / 768 / 1024 rows 36 / 50 / 57 us on tiles against 45 / 82 / 82 us mid-M; 4096 x 12288 at 256 / 768 rows
        # 50 / 83 against 58 / 125; 24576 x 4096 at 192 / 256 rows 67 / 69 against 92 / 99; but 4096 x 4096 at 192 / 256
        # rows 21 / 23 us mid-M against 27 / 28).  From 640 rows on, weights of 48 MB and more do best on plain 256 x 256
        # tiles even when those leave compute units idle (4096 x 12288 at 768 rows: 83 us against 100 for the plan's pick).
        # With the mid-M plan taking 128-column strips where K is long, everything below 6144 columns stays on it up to
        # 256 rows (5120 columns at 192 / 256 rows: 1.1 - 1.35 x against the tiles for K 1024 .. 27648), and 6144 .. 12288
        # columns do best on 256 x 128 tiles there (1.07 - 1.16 x against the plan's pick).
        # Short K at those widths is mid-M's as well (7680 x 2560 at 192 / 256 rows: 22 / 26 us against 29 / 31).
        narrow = M <= 256 and (N < 6144 or (N <= 8192 and K < 4096))    # (few column tiles: mid-M's K split fills the chip)
        # Further up it stays ahead where the tiles are fewest: up to 2048 columns at 512 rows (1.3 x), below 4096
        # columns at 384 rows (1.05 - 1.27 x), K up to 1280 at up to 512 rows and 8192 columns (1.15 - 1.4 x; at 768 rows
        # 7168 x 1024 takes 27 us there against 17 on tiles).  Of the tile shapes, 128 x 256 is ahead at 257 .. 384 rows
        # and 4096 .. 8192 columns (1.04 - 1.14 x against the plan's pick), 256 x 128 at 385 .. 512 rows and 2560 .. 7168
        # columns (1.07 - 1.17 x; K up to 8192, to 16384 up to 4096 columns) and at 513 .. 768 rows up to 3584 columns
        # (2560 x 9728: 51 us against 70).
        narrow = narrow or (M <= 512 and N <= 2048) or (M <= 384 and N < 4096) or (M <= 512 and N <= 8192 and K <= 1280)
        if M > 256 and not narrow and MID_BAND_TILES and gemm_tiles_supported(a, b) and not force_native:
            if M <= 384 and 4096 <= N <= 8192 and K <= 16384 and K % 2048 != 1024:      # (K = 5120, 7168: the plan's pick)
                return gemm_nt_tiles(a, b, out, bm=128, bn=256, group_m=group_m)
            if 384 < M <= 512 and 2560 <= N <= 7168 and 3072 <= K <= (16384 if N <= 4096 else 8192):
                return gemm_nt_tiles(a, b, out, bm=256, bn=128, group_m=group_m)
            if 512 < M <= 768 and N <= 3584 and K >= 4096:
                return gemm_nt_tiles(a, b, out, bm=256, bn=128, group_m=group_m)
        if (128 < M <= 256 and 6144 <= N < 16384 and not narrow and K >= 2048 and MID_BAND_TILES and gemm_tiles_supported(a, b)
                and not force_native):
            return gemm_nt_tiles(a, b, out, bm=256, bn=128, group_m=group_m)
        if M > 128 and not narrow and MID_BAND_TILES and gemm_tiles_supported(a, b) and not force_native:
            if M >= 640 and N * K >= 24_000_000:
                return gemm_nt_tiles(a, b, out, bm=256, bn=256, group_m=group_m)
            return gemm_nt_tiles(a, b, out, group_m=group_m)
        if M <= MID_MAX_M:
            return gemm_nt_mid(a, b, out)
    # Decode shapes (M <= 16): the weight-streaming kernels (csrc/ops/gemm.hip: one row on the vector ALUs, 2 .. 16 rows
    # on MFMAs with 128-byte runs per weight row) take everything except long-K layers above eight rows, where the
    # mid-M kernel (rows padded to 32, split-K) is ahead: 4096 x 12288 at 2 / 4 / 8 / 16 rows 20.6 / 21.3 / 23.5 / 28.8 us
    # streaming against 23.9 / 23.9 / 24.1 / 24.3 us mid-M; 6144 x 4096 12.6 .. 16.4 against 16.3 .. 16.7.
    # From 12288 columns on the streaming kernel with 64-column blocks and eight K slices is ahead again (12800 x 8192 at
    # 16 rows: 42 us against 48 us mid-M, cold weights).
    # TDIST_DECODE_SPLITK_MIN_M moves the first row count (17 = never); TDIST_DECODE_SPLITK=1 is 1.
    # Between 4096 and 6144 columns the streaming kernel's grid is a bad fit (16-column blocks: 256 .. 384 workgroups on
    # 256 compute units) and the mid-M kernel is ahead much earlier: 5120 columns, cold weights, 4 / 8 / 16 rows against
    # the best streaming shape 1.19 / 1.27 / 1.47 x at K = 5120 and 1.41 / 1.53 x at K = 25600 (4 / 8 rows).
    if M <= 16 and variant == GEMM_VARIANT and _decode_takes_mid(M, N, K):
        return gemm_nt_mid(a, b, out)
    # Otherwise decode shapes stay on the weight-streaming kernel.  Called back to back on ONE 100-270 MB layer the
    # library's 64x16x512 kernel looks faster (6.3-6.6 vs 4-5 TB/s), but there the weights partly sit in
    # the 256 MB memory-side cache; in a real decode step (Qwen3-8B, every layer streamed from HBM) the
    # step takes 4.51 ms with this kernel everywhere against 4.69 ms with those layers on the library
    # (profiles/decode_gemm_r1.md).  TDIST_DECODE_GEMM_LIB_MB sets a size from which the library is used.
    if (not force_native and DECODE_LIB_BYTES > 0 and M <= 16 and K >= 4096
            and N * K * a.element_size() >= DECODE_LIB_BYTES):
        torch.matmul(a, b.t(), out=out)
        return out
    # More than MID_MAX_M rows but fewer 256x256 tiles than ~60 % of the CUs (prefill of a mid-sized prompt
    # against a narrow projection: 2048 x 4096 = 128 tiles for 256 CUs): one 256x256 item per tile leaves CUs idle,
    # cutting K through parked fp32 tiles (gemm_nt_splitk) moves 64+ MB of partials and the mid-M kernel re-reads the
    # activations (profiles/gemm_few_tiles_r2.md), so this band runs on the tile family's smaller macro-tiles.
    # ``TDIST_FEW_TILES_GEMM`` = tiles (default) | lib (hipBLASLt) | native (256x256 tiles) | mid | splitk.
    tiles = -(-M // 256) * -(-N // 256)
    if M > 16 and variant == GEMM_VARIANT and tiles < 160 and not force_native and FEW_TILES_GEMM != "native":
        if FEW_TILES_GEMM == "mid":
            return gemm_nt_mid(a, b, out)
        if FEW_TILES_GEMM == "splitk" and K >= 1024:
            return gemm_nt_splitk(a, b, out)
        if FEW_TILES_GEMM == "lib":
            torch.matmul(a, b.t(), out=out)
            return out
        return gemm_nt_tiles(a, b, out, group_m=group_m)      # 256x128 / 128x256 / 128x128 tiles by cost model
    # More tiles than CUs but not a whole number of rounds (320 tiles = 1.25 rounds): the tile family's plan keeps the
    # whole rounds on 256x256 tiles and cuts only the last round's tiles along K so it fills the chip too
    # (csrc/include/tdist/tile_plan.hpp), or picks another macro-tile when that is cheaper.
    if (M > 16 and variant == GEMM_VARIANT and not force_native and FEW_TILES_GEMM == "tiles"
            and tiles > _num_cus(a.device) and tiles % _num_cus(a.device)):
        return gemm_nt_tiles(a, b, out, group_m=group_m)
    lib = _lib.hip()
    _lib.check(
        lib.tdist_gemm_nt(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), M, N, K, ctypes.c_longlong(a.stride(0)),
                          ctypes.c_longlong(b.stride(0)), ctypes.c_longlong(out.stride(0)), _DT[a.dtype], group_m,
                          _lib.stream_handle(), variant), "tdist_gemm_nt")
    return out


FEW_TILES_GEMM = os.environ.get("TDIST_FEW_TILES_GEMM", "tiles")

# operand types of the tile family (csrc/include/tdist/gemm_gfx950.hpp: DT_*; C ABI numbering)
_DT_EX = {torch.bfloat16: 0, torch.float16: 1, torch.float8_e4m3fn: 2, torch.float8_e5m2: 3, torch.int8: 4}


def gemm_tiles_supported(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Shapes / types ``gemm_nt_tiles`` takes: K-contiguous 2-D operands of one of the five operand types, K a
    multiple of 64 (16-bit) or 128 (8-bit) and at least two such steps, N a multiple of 8, 16-byte aligned rows."""
    if not (a.is_cuda and b.is_cuda) or a.dtype not in _DT_EX or a.dtype != b.dtype:
        return False
    if a.dim() != 2 or b.dim() != 2 or a.stride(1) != 1 or b.stride(1) != 1:
        return False
    M, K = a.shape
    N = b.shape[0]
    per16 = 2 // a.element_size()
    return (M > 0 and N > 0 and N % 8 == 0 and K % (64 * per16) == 0 and K >= 128 * per16
            and a.stride(0) % (8 * per16) == 0 and b.stride(0) % (8 * per16) == 0
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)


def gemm_nt_tiles(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None,
                  scale_a: Optional[torch.Tensor] = None, scale_b: Optional[torch.Tensor] = None,
                  out_dtype: Optional[torch.dtype] = None, bm: int = 0, bn: int = 0, split_k: int = 0,
                  group_m: Optional[int] = None, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``a @ b^T`` on the tile family (csrc/ops/gemm.hip: gemm_nt_tile_kernel): 256x256, 256x128, 128x256 or 128x128
    macro-tiles (``bm`` / ``bn``; 0 = cost model), optionally with K cut into ``split_k`` ranges joined wait-free
    inside the launch (0 = cost model), bf16 / fp16 / fp8 (OCP e4m3, e5m2) / int8 operands.

    Output: the input type for 16-bit inputs; bf16 for 8-bit inputs (``out_dtype`` None or bf16), optionally
    multiplied by ``scale_a`` [M] (per row of a) and ``scale_b`` [N] (per row of b) fp32 scales in the epilogue -
    the de-quantisation of per-token / per-channel quantised operands; ``out_dtype=torch.int32`` returns the exact
    int32 accumulators of an int8 GEMM.  fp8 runs on the 128-deep MFMA (twice the bf16 rate), int8 on the 64-deep one.
    ``bias`` [N] of the output type is added to the 16-bit output in the same epilogue.

    Role in the reference: the dtype-generic consumer GEMM of kernels/amd/all_gather_gemm.py:127-245 (``--dtype
    float8_e4m3fn | float8_e5m2 | s8`` of test/amd/test_ag_gemm_intra_node.py:122-136) and its BM / BN autotune space."""
    assert a.shape[1] == b.shape[1], f"K mismatch: {tuple(a.shape)} vs {tuple(b.shape)}"
    if not gemm_tiles_supported(a, b):
        raise ValueError("shape / dtype not supported by the gfx950 tile family")
    M, K = a.shape
    N = b.shape[0]
    eight = a.element_size() == 1
    if out_dtype is None:
        out_dtype = torch.bfloat16 if eight else a.dtype
    want32 = out_dtype == torch.int32
    if want32 and a.dtype != torch.int8:
        raise ValueError("int32 output needs int8 operands")
    if not want32 and out_dtype != (torch.bfloat16 if eight else a.dtype):
        raise ValueError(f"unsupported output type {out_dtype} for {a.dtype} operands")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.dtype == out_dtype and out.shape == (M, N) and out.stride(1) == 1
    assert out.stride(0) % (4 if want32 else 8) == 0 and out.data_ptr() % 16 == 0
    for sc, n in ((scale_a, M), (scale_b, N)):
        assert sc is None or (sc.dtype == torch.float32 and sc.is_contiguous() and sc.numel() == n and not want32)
    assert bias is None or (bias.dtype == out_dtype and bias.is_contiguous() and bias.numel() == N and not want32)
    if group_m is None:
        group_m = tuning.get("gemm.group_m", 8)
    lib = _lib.hip()
    pbm, pbn, ps = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    pbytes, ncnt, nfull = ctypes.c_longlong(), ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.tdist_gemm_nt_ex_plan(M, N, K, _DT_EX[a.dtype], bm, bn, split_k or tuning.get("gemm.TILE_SPLIT_K", 0),
                                         ctypes.byref(pbm), ctypes.byref(pbn), ctypes.byref(ps), ctypes.byref(pbytes),
                                         ctypes.byref(ncnt), ctypes.byref(nfull)), "tdist_gemm_nt_ex_plan")
    ws = _tile_workspace(a.device, pbytes.value, ncnt.value) if ps.value > 1 else None
    _lib.check(
        lib.tdist_gemm_nt_ex_bias(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), M, N, K, ctypes.c_longlong(a.stride(0)),
                                  ctypes.c_longlong(b.stride(0)), ctypes.c_longlong(out.stride(0)), _DT_EX[a.dtype],
                                  1 if want32 else 0, _lib.ptr(scale_a) if scale_a is not None else ctypes.c_void_p(0),
                                  _lib.ptr(scale_b) if scale_b is not None else ctypes.c_void_p(0),
                                  _lib.ptr(bias) if bias is not None else ctypes.c_void_p(0), pbm.value, pbn.value,
                                  ps.value, _lib.ptr(ws[0]) if ws else ctypes.c_void_p(0),
                                  ctypes.c_longlong(ws[0].numel() if ws else 0),
                                  _lib.ptr(ws[1]) if ws else ctypes.c_void_p(0), ws[1].numel() if ws else 0, group_m,
                                  _lib.stream_handle()), "tdist_gemm_nt_ex_bias")
    return out


_TILE_WS = {}     # (device index, stream) or (device index, "capture", id) -> (fp32 partial tiles, arrival counters)


def _tile_workspace(device, pbytes: int, ncnt: int):
    """Split-K workspace of ``gemm_nt_tiles``: one per device and stream (launches on a stream run in order and each
    leaves its counters zero); under stream capture one per capture, from the graph's pool (as ``gemm_nt_mid``)."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    if torch.cuda.is_current_stream_capturing():
        cid = ctypes.c_ulonglong()
        _lib.check(_lib.hip().tdist_stream_capture_id(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                                                      ctypes.byref(cid)), "stream_capture_id")
        key = (dev, "capture", cid.value)
        for k in [k for k in _TILE_WS if len(k) == 3 and k[0] == dev and k[2] != cid.value]:
            del _TILE_WS[k]
    else:
        key = (dev, torch.cuda.current_stream().cuda_stream)
    ws = _TILE_WS.get(key)
    if ws is None or ws[0].numel() < pbytes or ws[1].numel() < ncnt:
        if ws is not None and len(key) == 3:
            raise RuntimeError("gemm_nt_tiles: a later call of this capture needs a larger split-K workspace than the "
                               "first one allocated; run the largest shape first")
        ws = (torch.empty(max(pbytes, 32 << 20, ws[0].numel() if ws else 0), dtype=torch.uint8, device=device),
              torch.zeros(max(ncnt, 4096, ws[1].numel() if ws else 0), dtype=torch.int32, device=device))
        _TILE_WS[key] = ws
    return ws


_CUS = {}


def _num_cus(device) -> int:
    dev = device.index if device.index is not None else torch.cuda.current_device()
    if dev not in _CUS:
        _CUS[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    return _CUS[dev]


def gemm_nt_splitk(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, split_k: int = 0) -> torch.Tensor:
    """``a @ b^T`` on 256x256 tiles with K cut into ``split_k`` ranges (default: as many as fill the chip, at most
    4, each >= 512 deep), joined wait-free inside the launch (``gemm_nt_tiles`` with the tile pinned).  The kernel
    re-arms its own counters, so the launch is one kernel node (HIP-graph capturable) and consecutive calls on a
    stream share one workspace."""
    M, K = a.shape
    N = b.shape[0]
    cus = _num_cus(a.device)
    tiles = -(-M // 256) * -(-N // 256)
    s = split_k or tuning.get("gemm.SPLIT_K", 0) or max(1, min(4, cus // tiles, K // 512))
    s = max(1, min(s, 4, K // 128))
    return gemm_nt_tiles(a, b, out, bm=256, bn=256, split_k=s)


_MID_WS = {}      # device index -> (fp32 split-K workspace, arrival counters)


def gemm_nt_mid(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, wn: int = 0,
                splits: int = 0) -> torch.Tensor:
    """``a @ b^T`` on the mid-M kernel (csrc/ops/gemm_mid.hip): weights streamed once, activations through
    LDS, K split across workgroups.  ``wn`` (1 / 2: 64 / 128 columns per workgroup) and ``splits`` override the
    plan (tuning knobs ``gemm_mid.WN`` / ``gemm_mid.SPLITS``)."""
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    lib = _lib.hip()
    dev = a.device.index if a.device.index is not None else torch.cuda.current_device()
    mt, wn_, sp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    pbytes, ncnt = ctypes.c_longlong(), ctypes.c_int()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    _lib.check(lib.tdist_gemm_nt_mid_plan(M, N, K, cus, wn or tuning.get("gemm_mid.WN", 0),
                                          splits or tuning.get("gemm_mid.SPLITS", 0), ctypes.byref(mt),
                                          ctypes.byref(wn_), ctypes.byref(sp), ctypes.byref(pbytes),
                                          ctypes.byref(ncnt)), "tdist_gemm_nt_mid_plan")
    # One workspace per device and STREAM: launches on one stream run in order and every launch leaves
    # its counters zero, so consecutive GEMMs can share it; concurrent streams must not.
    if torch.cuda.is_current_stream_capturing():
        # Capture runs on a stream of its own: the workspace comes from the graph's private pool (it lives as
        # long as the graph does) and is shared by every call recorded into the same capture, so a graph
        # replays ONE zero-fill node for all its split-K launches.  Keyed by the capture's identity: a later
        # capture never picks up memory of a graph that may be gone.
        cid = ctypes.c_ulonglong()
        _lib.check(lib.tdist_stream_capture_id(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                                                 ctypes.byref(cid)), "stream_capture_id")
        key = (dev, "capture", cid.value)
        ws = _MID_WS.get(key)
        if ws is None or ws[0].numel() < pbytes.value or ws[1].numel() < ncnt.value:
            for k in [k for k in _MID_WS if len(k) == 3 and k[0] == dev and k[2] != cid.value]:
                del _MID_WS[k]                       # workspaces of finished captures: the graphs hold their memory
            ws = (torch.empty(max(pbytes.value, 8 << 20), dtype=torch.uint8, device=a.device),
                  torch.zeros(max(ncnt.value, 4096), dtype=torch.int32, device=a.device))
            _MID_WS[key] = ws
    else:
        key = (dev, torch.cuda.current_stream().cuda_stream)
        ws = _MID_WS.get(key)
        if ws is None or ws[0].numel() < pbytes.value or ws[1].numel() < ncnt.value:
            old = ws
            ws = (torch.empty(max(pbytes.value, 1 << 20, old[0].numel() if old else 0), dtype=torch.uint8,
                              device=a.device),
                  torch.zeros(max(ncnt.value, 4096, old[1].numel() if old else 0), dtype=torch.int32, device=a.device))
            _MID_WS[key] = ws
    _lib.check(
        lib.tdist_gemm_nt_mid(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), M, N, K, ctypes.c_longlong(a.stride(0)),
                              ctypes.c_longlong(b.stride(0)), ctypes.c_longlong(out.stride(0)), _DT[a.dtype],
                              mt.value, wn_.value, sp.value, _lib.ptr(ws[0]), _lib.ptr(ws[1]),
                              _lib.stream_handle()), "tdist_gemm_nt_mid")
    return out


def gemm_nt_silu_mul(a: torch.Tensor, gate_up: torch.Tensor) -> torch.Tensor:
    """``silu(a @ gate^T) * (a @ up^T)`` for ``gate_up = [gate; up]`` of shape [2 I, K] -> [M, I].

    M <= 16 in bf16 runs ONE launch (the decode GEMM with the activation in its epilogue); everything
    else is ``gemm_nt`` followed by the activation kernel."""
    M, K = a.shape
    I = gate_up.shape[0] // 2
    if (M <= 16 and a.dtype == torch.bfloat16 and gemm_supported(a, gate_up) and gate_up.shape[0] == 2 * I
            and I % 2 == 0 and K % 32 == 0):
        out = torch.empty((M, I), dtype=a.dtype, device=a.device)
        _lib.check(
            _lib.hip().tdist_gemm_nt_silu_mul(_lib.ptr(a), _lib.ptr(gate_up), _lib.ptr(out), M, I, K,
                                              ctypes.c_longlong(a.stride(0)), ctypes.c_longlong(gate_up.stride(0)),
                                              ctypes.c_longlong(out.stride(0)), _lib.stream_handle()),
            "tdist_gemm_nt_silu_mul")
        return out
    from .elementwise import silu_mul
    return silu_mul(gemm_nt(a, gate_up))


# Rows up to which the norm is folded into the decode GEMM.  Every workgroup of the fused kernel normalises all rows
# again, which pays for one and two rows only: 12 layers of Qwen3-4B at batch 2 / 3 / 4 / 8 take 1.025 / 1.069 / 1.123 /
# 1.366 ms fused against 1.044 / 1.055 / 1.074 / 1.119 ms with the norm as its own launch, Qwen3-8B at batch 3 / 4
# 1.330 / 1.379 against 1.341 / 1.364, Qwen3-14B at batch 4 2.197 against 2.161 (TDIST_NORM_GEMM_MAX_M, at most 16).
NORM_GEMM_MAX_M = min(16, int(os.environ.get("TDIST_NORM_GEMM_MAX_M", "2")))


def norm_gemm_supported(x: torch.Tensor, w: torch.Tensor, norm_w: torch.Tensor, residual: Optional[torch.Tensor]) -> bool:
    """Shapes the decode GEMM takes with the RMSNorm folded in: at most NORM_GEMM_MAX_M rows (the kernel itself takes
    16), bf16, the rows' normalised copy (M x (2 K + 64) bytes) within 48 KiB of LDS."""
    M, K = x.shape
    return (x.is_cuda and M <= NORM_GEMM_MAX_M and x.dtype == torch.bfloat16 == w.dtype == norm_w.dtype and gemm_supported(x, w)
            and K % 32 == 0 and K >= 128 and M * (2 * K + 64) <= 48 * 1024 and norm_w.is_contiguous()
            and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape)))


def gemm_nt_norm(x: torch.Tensor, norm_w: torch.Tensor, eps: float, w: torch.Tensor,
                 residual: Optional[torch.Tensor] = None, silu: bool = False):
    """``rmsnorm(x [+ residual]) * norm_w @ w^T`` -> (out, hidden) with hidden = x + residual (or x).
    ``silu``: ``w`` is a gate_up weight [2 I, K] and out = silu(gate) * up [M, I].

    Decode-sized bf16 inputs run ONE launch (the weight-streaming GEMM computes the norm from the fragments it
    loads anyway: csrc/ops/gemm.hip, NORM); everything else is the norm kernel followed by the GEMM.  In the fused
    form the MFMAs see bf16(h * norm_w) and the row scale is applied in fp32 afterwards, so results differ from the
    two-kernel form by bf16 rounding order only."""
    from .elementwise import rms_norm

    M, K = x.shape
    N = w.shape[0] // (2 if silu else 1)
    if norm_gemm_supported(x, w, norm_w, residual) and N % 2 == 0 and x.stride(1) == 1:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
        res = residual.contiguous() if residual is not None else None
        hidden = torch.empty((M, K), dtype=x.dtype, device=x.device) if res is not None else x
        _lib.check(
            _lib.hip().tdist_gemm_nt_norm(_lib.ptr(x), _lib.ptr(res), _lib.ptr(norm_w), ctypes.c_float(eps), _lib.ptr(w),
                                          _lib.ptr(out), _lib.ptr(hidden) if res is not None else ctypes.c_void_p(0), M, N, K,
                                          ctypes.c_longlong(x.stride(0)), ctypes.c_longlong(w.stride(0)),
                                          ctypes.c_longlong(out.stride(0)), 1 if silu else 0, _lib.stream_handle()),
            "tdist_gemm_nt_norm")
        return out, hidden
    if residual is not None:
        y, hidden = rms_norm(x, norm_w, eps, residual=residual)
    else:
        y, hidden = rms_norm(x, norm_w, eps), x
    return (gemm_nt_silu_mul(y, w) if silu else gemm_nt(y, w)), hidden
