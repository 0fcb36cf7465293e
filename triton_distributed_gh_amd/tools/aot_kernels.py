"""The in-tree AOT kernel set: specs + native launchers.

Mirrors the reference's ``scripts/aot_kernels.txt`` (five flash-decode kernels + two test kernels)
and its AOT call sites (kernels/nvidia/flash_decode.py:70-128 specs, :979-1132 hosts): every entry
is callable as ``distributed.<name>(stream, raw pointers..., scalars..., algo_info)``.  The
launchers call the gfx950 kernels of libtdist_hip.so with those raw pointers; dtypes cover bf16 and
fp16 inputs, and the split results are fp32 (what this framework's own layers use) or, as in the reference's
``*_fp16_fp16_fp16`` / ``*_combine_kv_fp16_fp16`` variants (flash_decode.py:112-128, :289-305, :364-385,
:453-474, :578-586), the 16-bit type of the inputs."""
from __future__ import annotations

import ctypes

from .. import _lib
from .compile_aot import aot_compile_spaces

_LL = ctypes.c_longlong
_P = ctypes.c_void_p
_DT = {"bf16": 0, "fp16": 1}


def get_split_kv_algo_info(q_heads, kv_heads, q_head_dim, v_head_dim, page_size, split_kv=32, soft_cap=0.0):
    pow2 = 1 << (int(q_head_dim).bit_length() - 1)
    return {"kv_group_num": q_heads // kv_heads, "q_head_num": q_heads, "BLOCK_HEAD_DIM": pow2,
            "BLOCK_DPE": q_head_dim - pow2, "BLOCK_DV": 1 << (int(v_head_dim) - 1).bit_length(), "BLOCK_N": 64,
            "BLOCK_H": 16, "NUM_KV_SPLITS": split_kv, "PAGE_SIZE": page_size, "soft_cap": soft_cap, "K_DIM": q_head_dim,
            "V_DIM": v_head_dim, "num_warps": 4, "num_stages": 2}


def get_combine_kv_algo_info(split_kv=32, v_head_dim=128):
    return {"BLOCK_DV": 1 << (int(v_head_dim) - 1).bit_length(), "NUM_KV_SPLITS": split_kv, "V_DIM": v_head_dim,
            "num_warps": 4, "num_stages": 2}


_SPLIT_SIG = ("*{i}:16, *{c}:16, *{c}:16, *{o}:16, fp32, *i32:16, *i32, i32, "
              + ", ".join(["i32:16"] * 8 + ["i32", "i32"]) + ", "
              "%kv_group_num, %q_head_num, %BLOCK_HEAD_DIM, %BLOCK_DPE, %BLOCK_DV, %BLOCK_N, %BLOCK_H, %NUM_KV_SPLITS, "
              "%PAGE_SIZE, %soft_cap, %K_DIM, %V_DIM")
_SPLIT_GRID = ["batch",
               "(%q_head_num + (%BLOCK_H < %kv_group_num ? %BLOCK_H : %kv_group_num) - 1) / "
               "(%BLOCK_H < %kv_group_num ? %BLOCK_H : %kv_group_num)", "%NUM_KV_SPLITS"]
_SPLIT_ALGOS = [get_split_kv_algo_info(96, 12, 128, 128, 1), get_split_kv_algo_info(24, 3, 128, 128, 1),
                get_split_kv_algo_info(64, 8, 128, 128, 64), get_split_kv_algo_info(32, 8, 128, 128, 16)]


def _name_dtype(name: str) -> int:
    """dtype code of a kernel name ``..._<in>_<cache>_<out>`` / ``..._<in>_<out>``: the 16-bit type named in it
    (bf16 0, fp16 1), + 16 when the partial rows are 16-bit too (no ``fp32`` in the name)."""
    tail = name.replace("persistent_", "").split("_kv_")[-1].split("_") if "_kv_" in name else []
    code = 1 if "fp16" in tail else 0
    return code + (0 if "fp32" in tail or not tail else 16)


def _split_launch(stream, q, k, v, out, sm_scale, table, kv_len, batch, qs0, qs1, ks_slot, ks_h, vs_slot, vs_h, os0, os1,
                  os2, ts0, info, c_kernel_name="", fused=0, final=None, final_strides=(0, 0), tickets=None):
    D, splits = int(info["K_DIM"]), int(info["NUM_KV_SPLITS"])
    heads = int(info["q_head_num"])
    kv_heads = heads // int(info["kv_group_num"])
    if int(info["V_DIM"]) != D:
        raise ValueError("the native decode kernel needs equal K and V head dims")
    if (os2, os1, os0) != (D + 1, splits * (D + 1), heads * splits * (D + 1)):
        raise ValueError("output_split must be a contiguous fp32 [batch, q_heads, splits, V_DIM + 1] tensor")
    dtype = _name_dtype(c_kernel_name)
    _lib.check(
        _lib.hip().tdist_flash_decode_split(
            _P(q), _P(k), _P(v), _P(table), _P(kv_len), _P(out), _P(final), _P(tickets), int(batch), heads, kv_heads, D,
            splits, int(info["PAGE_SIZE"]), int(ts0), _LL(qs0), _LL(qs1), _LL(ks_slot), _LL(ks_h), _LL(vs_slot),
            _LL(vs_h), _LL(final_strides[0]), _LL(final_strides[1]), ctypes.c_float(sm_scale),
            ctypes.c_float(float(info.get("soft_cap", 0.0))), fused, dtype, _P(stream)), c_kernel_name)


@aot_compile_spaces({
    f"gqa_fwd_batch_decode_split_kv_{t}_{t}_{o}": {"signature": _SPLIT_SIG.format(i=t, c=t, o=o),
                                                    "grid": _SPLIT_GRID, "triton_algo_infos": _SPLIT_ALGOS}
    for t in ("bf16", "fp16") for o in ("fp32", t)})
def kernel_gqa_fwd_batch_decode_split_kv(stream, *args, c_kernel_name=""):
    """Split-KV pass: output_split [batch, q_heads, splits, V_DIM + 1] (normalised rows, LSE last) in fp32 or
    in the input's 16-bit type, as the kernel name says."""
    _split_launch(stream, *args, c_kernel_name=c_kernel_name)


_COMBINE_SIG = "*{i}:16, *{o}:16, *i32, i32, i32, i32:16, i32:16, i32, i32:16, i32:16, %BLOCK_DV, %NUM_KV_SPLITS, %V_DIM"
_COMBINE_ALGOS = [get_combine_kv_algo_info(32, 128), get_combine_kv_algo_info(16, 128), get_combine_kv_algo_info(8, 128)]


def _combine(stream, part, out, lens, batch, heads, ps_b, ps_h, ps_s, os_b, os_h, info, out_mode, name, n=None,
             ps_n=None, lens_stride=0):
    dtype = _name_dtype(name)
    _lib.check(
        _lib.hip().tdist_flash_decode_combine(_P(part), _LL(ps_s if ps_n is None else ps_n), _LL(ps_b), _LL(ps_h),
                                              int(info["NUM_KV_SPLITS"]) if n is None else int(n), int(batch), int(heads),
                                              int(info["V_DIM"]), _P(lens), int(lens_stride), _P(out), _LL(os_b), _LL(os_h),
                                              out_mode, dtype, _P(stream)), name)


@aot_compile_spaces({
    f"gqa_fwd_batch_decode_combine_kv_{i}_{t}": {"signature": _COMBINE_SIG.format(i=i, o=t),
                                                 "grid": ["batch", "q_head_num", "1"],
                                                 "triton_algo_infos": _COMBINE_ALGOS}
    for t in ("bf16", "fp16") for i in ("fp32", t)})
def kernel_gqa_fwd_batch_decode_combine_kv(stream, part, out, kv_len, batch, q_head_num, ps_b, ps_h, ps_s, os_b, os_h, info,
                                           c_kernel_name=""):
    """partial [batch, heads, splits, V_DIM + 1] (fp32 or 16-bit) -> out [batch, heads, V_DIM] (16-bit)."""
    _combine(stream, part, out, None, batch, q_head_num, ps_b, ps_h, ps_s, os_b, os_h, info, 1, c_kernel_name)


@aot_compile_spaces({
    **{"intra_rank_gqa_fwd_batch_decode_combine_kv_fp32_fp32": {
        "signature": _COMBINE_SIG.format(i="fp32", o="fp32"), "grid": ["batch", "q_head_num", "1"],
        "triton_algo_infos": _COMBINE_ALGOS}},
    **{f"intra_rank_gqa_fwd_batch_decode_combine_kv_{i}_{t}": {
        "signature": _COMBINE_SIG.format(i=i, o=t), "grid": ["batch", "q_head_num", "1"],
        "triton_algo_infos": _COMBINE_ALGOS} for t in ("bf16", "fp16") for i in ("fp32", t)}})
def kernel_intra_rank_gqa_fwd_batch_decode_combine_kv(stream, part, out, kv_len, batch, q_head_num, ps_b, ps_h, ps_s, os_b,
                                                      os_h, info, c_kernel_name=""):
    """Like the combine, but emits [batch, heads, V_DIM + 1] with the LSE in the last column: fp32
    (``..._fp32_fp32``, what this framework's SP decode layer exchanges between ranks) or 16-bit."""
    out_mode = 2 if c_kernel_name.endswith("_fp32") else 3
    _combine(stream, part, out, None, batch, q_head_num, ps_b, ps_h, ps_s, os_b, os_h, info, out_mode, c_kernel_name)


_INTER_SIG = "*{i}:16, *{o}:16, *i32, i32, i32, i32, i32:16, i32:16, i32:16, i32:16, i32:16, %BLOCK_DV, %NUM_KV_SPLITS, %V_DIM"


@aot_compile_spaces({
    f"inter_rank_gqa_fwd_batch_decode_combine_kv_{i}_{t}": {
        "signature": _INTER_SIG.format(i=i, o=t), "grid": ["batch", "q_head_num", "1"],
        "triton_algo_infos": _COMBINE_ALGOS} for t in ("bf16", "fp16") for i in ("fp32", t)})
def kernel_inter_rank_gqa_fwd_batch_decode_combine_kv(stream, part, out, kv_lens, num_ranks, batch, q_head_num, ps_r, ps_b,
                                                      ps_h, os_b, os_h, info, c_kernel_name=""):
    """partials [num_ranks, batch, heads, V_DIM + 1] (+ kv_lens [num_ranks, batch]) -> out 16-bit;
    ranks whose KV length for a sequence is 0 are skipped."""
    _combine(stream, part, out, kv_lens, batch, q_head_num, ps_b, ps_h, 0, os_b, os_h, info, 1, c_kernel_name,
             n=num_ranks, ps_n=ps_r, lens_stride=batch)


_PERSISTENT_SIG = ("*{i}:16, *{c}:16, *{c}:16, *{o}:16, *{o}:16, fp32, *i32:16, *i32, *i32:16, i32, "
                   + ", ".join(["i32:16"] * 8 + ["i32"] + ["i32:16"] * 2 + ["i32"]) + ", "
                   "%kv_group_num, %max_kv_seq_len, %q_head_num, %BLOCK_HEAD_DIM, %BLOCK_DPE, %BLOCK_DV, %BLOCK_N, %BLOCK_H, "
                   "%NUM_KV_SPLITS, %PAGE_SIZE, %soft_cap, %K_DIM, %V_DIM")


def get_persistent_algo_info(q_heads, kv_heads, q_head_dim, v_head_dim, page_size, split_kv=32, soft_cap=0.0,
                             max_kv_len=8192):
    info = get_split_kv_algo_info(q_heads, kv_heads, q_head_dim, v_head_dim, page_size, split_kv, soft_cap)
    info.update({"max_kv_seq_len": max_kv_len, "num_warps": 8})
    return info


@aot_compile_spaces({
    f"gqa_fwd_batch_decode_split_kv_persistent_{t}_{t}_{t}": {
        "signature": _PERSISTENT_SIG.format(i=t, c=t, o=t), "grid": ["256", "1", "1"],
        "triton_algo_infos": [get_persistent_algo_info(96, 12, 128, 128, 1), get_persistent_algo_info(64, 8, 128, 128, 64)]}
    for t in ("bf16", "fp16")})
def kernel_gqa_fwd_batch_decode_split_kv_persistent(stream, q, k, v, out_split, final_out, sm_scale, table, kv_len,
                                                    workspace, batch, qs0, qs1, ks_slot, ks_h, vs_slot, vs_h, os0, os1, os2,
                                                    fs0, fs1, ts0, info, c_kernel_name=""):
    """Single launch (argument list of the reference's persistent kernel, flash_decode.py:587-631): split-KV
    pass into ``out_split`` ([batch, q_heads, splits, V_DIM + 1], the 16-bit type) and the merge into
    ``final_out`` ([batch, q_heads, V_DIM]) by the last workgroup to finish each (batch, head block).
    ``workspace``: int32, at least batch x head blocks entries, zero before the first launch (every launch
    leaves it zero); here it counts arrivals per head block instead of carrying a grid barrier."""
    _split_launch(stream, q, k, v, out_split, sm_scale, table, kv_len, batch, qs0, qs1, ks_slot, ks_h, vs_slot, vs_h, os0,
                  os1, os2, ts0, info, c_kernel_name=c_kernel_name, fused=1, final=final_out, final_strides=(fs0, fs1),
                  tickets=workspace)


# ---- the reference's two test kernels (test/nvidia/test_compile_aot.py) ------------------------------
@aot_compile_spaces({
    "test_aot_kernel_bf16": {"signature": "*bf16:16, *bf16:16, *bf16:16, i32, %BLOCK_SIZE", "grid": ["(n + %BLOCK_SIZE - 1) / %BLOCK_SIZE", "1", "1"],
                             "triton_algo_infos": [{"BLOCK_SIZE": 1024, "num_warps": 4, "num_stages": 1},
                                                   {"BLOCK_SIZE": 2048, "num_warps": 8, "num_stages": 1}]}})
def test_aot_kernel(stream, a, b, y, n, info, c_kernel_name=""):
    """y = a + b over n bf16 elements."""
    _lib.check(_lib.hip().tdist_add(_P(a), _P(b), _P(y), _LL(n), _P(stream)), c_kernel_name)


test_aot_kernel.__test__ = False


@aot_compile_spaces({
    "matmul_kernel_descriptor_persistent_bf16": {
        "signature": "*bf16:16, *bf16:16, *bf16:16, i32, i32, i32, %BLOCK_SIZE_M, %BLOCK_SIZE_N, %BLOCK_SIZE_K, %GROUP_SIZE_M",
        "grid": ["((M + %BLOCK_SIZE_M - 1) / %BLOCK_SIZE_M) * ((N + %BLOCK_SIZE_N - 1) / %BLOCK_SIZE_N)", "1", "1"],
        "triton_algo_infos": [{"BLOCK_SIZE_M": 256, "BLOCK_SIZE_N": 256, "BLOCK_SIZE_K": 64, "GROUP_SIZE_M": g,
                               "num_warps": 8, "num_stages": 2} for g in (4, 8, 16)]}})
def matmul_kernel_descriptor_persistent(stream, a, b, c, M, N, K, info, c_kernel_name=""):
    """C[M, N] = A[M, K] @ B[N, K]^T (row-major, K contiguous) on the 256x256 MFMA tile engine."""
    _lib.check(_lib.hip().tdist_gemm_nt(_P(a), _P(b), _P(c), int(M), int(N), int(K), _LL(K), _LL(K), _LL(N), 0,
                                        int(info["GROUP_SIZE_M"]), _P(stream), 2), c_kernel_name)
