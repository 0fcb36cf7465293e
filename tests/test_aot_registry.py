"""AOT registry: spec validation, native registry round trip, algo selection, CLI manifest, and (GPU)
the reference's calling convention ``distributed.<kernel>(stream, args..., algo_info)``
(reference: python/triton_dist/test/nvidia/test_compile_aot.py:162-236)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from triton_distributed_gh_amd.tools import (distributed, parse_signature, registered_kernels, select_algo, validate_spec)
from triton_distributed_gh_amd.tools import aot_kernels  # noqa: F401  (registers the in-tree set)
from triton_distributed_gh_amd.tools.compile_aot import SpecError, eval_grid

ROOT = Path(__file__).resolve().parent.parent


def test_signature_parsing_and_validation():
    args = parse_signature("*fp16:16, *fp32, i32:1, i32:16, fp32, %BLOCK")
    assert [a["kind"] for a in args] == ["ptr", "ptr", "scalar", "scalar", "scalar", "const"]
    assert args[0]["hint"] == 16 and args[2]["hint"] == 1 and args[5]["name"] == "BLOCK"
    with pytest.raises(SpecError):
        parse_signature("*fp16:3")
    with pytest.raises(SpecError):
        validate_spec("k", {"signature": "i32, %B", "grid": ["1", "1", "1"], "triton_algo_infos": [{"num_warps": 4}]})
    with pytest.raises(SpecError):
        validate_spec("k", {"signature": "i32", "grid": ["%X", "1", "1"], "triton_algo_infos": [{"num_warps": 4}]})


def test_native_registry_lists_the_aot_kernel_set():
    names = registered_kernels()
    for n in ("gqa_fwd_batch_decode_split_kv_bf16_bf16_fp32", "gqa_fwd_batch_decode_combine_kv_fp32_bf16",
              "gqa_fwd_batch_decode_split_kv_fp16_fp16_fp16", "gqa_fwd_batch_decode_combine_kv_fp16_fp16",
              "intra_rank_gqa_fwd_batch_decode_combine_kv_fp16_fp16", "inter_rank_gqa_fwd_batch_decode_combine_kv_fp16_fp16",
              "gqa_fwd_batch_decode_split_kv_persistent_fp16_fp16_fp16",
              "intra_rank_gqa_fwd_batch_decode_combine_kv_fp32_fp32", "inter_rank_gqa_fwd_batch_decode_combine_kv_fp32_bf16",
              "test_aot_kernel_bf16", "matmul_kernel_descriptor_persistent_bf16"):
        assert n in names
        assert hasattr(distributed, n) and hasattr(distributed, n + "__triton_algo_info_t")
    info = aot_kernels.get_split_kv_algo_info(24, 3, 128, 128, 1)
    assert select_algo("gqa_fwd_batch_decode_split_kv_bf16_bf16_fp32", info) == 1
    assert select_algo("gqa_fwd_batch_decode_split_kv_bf16_bf16_fp32", {**info, "q_head_num": 7}) == -1
    spec = aot_kernels.kernel_gqa_fwd_batch_decode_split_kv.__aot_compile_spaces__["gqa_fwd_batch_decode_split_kv_bf16_bf16_fp32"]
    assert eval_grid(spec, info, {"batch": 5}) == (5, 3, 32)


def test_algo_info_struct_rejects_unknown_fields():
    t = distributed.test_aot_kernel_bf16__triton_algo_info_t()
    t.BLOCK_SIZE = 1024
    with pytest.raises(AttributeError):
        t.NOT_A_FIELD = 1


def test_cli_writes_manifest_and_builds_registration_library(tmp_path):
    ws = tmp_path / "aot"
    cmd = [sys.executable, "-m", "triton_distributed_gh_amd.tools.compile_aot", "--workspace", str(ws), "--from-scratch",
           "--kernels", f"{ROOT}/triton_distributed_gh_amd/tools/aot_kernels.py:test_aot_kernel",
           f"{ROOT}/triton_distributed_gh_amd/tools/aot_kernels.py:kernel_gqa_fwd_batch_decode_split_kv",
           "--library", "triton_distributed_kernel", "--build"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    man = json.loads((ws / "manifest.json").read_text())
    assert "test_aot_kernel_bf16" in man["kernels"] and len(man["kernels"]) == 5
    assert (ws / "libtriton_distributed_kernel.so").exists()


@pytest.mark.gpu
def test_aot_calling_convention_on_gpu():
    s = torch.cuda.current_stream().cuda_stream
    a = torch.randn(4096, device="cuda").bfloat16()
    b = torch.randn(4096, device="cuda").bfloat16()
    y = torch.empty_like(a)
    info = distributed.test_aot_kernel_bf16__triton_algo_info_t()
    info.BLOCK_SIZE, info.num_warps, info.num_stages = 1024, 4, 1
    distributed.test_aot_kernel_bf16(s, a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), info)
    assert torch.equal(y, a + b)
    info.BLOCK_SIZE = 333
    with pytest.raises(RuntimeError):
        distributed.test_aot_kernel_bf16(s, a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), info)
    # matmul entry vs torch
    M, N, K = 512, 768, 256
    A = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
    B = (torch.rand(N, K, device="cuda") - 0.5).bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    mi = distributed.matmul_kernel_descriptor_persistent_bf16__triton_algo_info_t(
        BLOCK_SIZE_M=256, BLOCK_SIZE_N=256, BLOCK_SIZE_K=64, GROUP_SIZE_M=8, num_warps=8, num_stages=2)
    distributed.matmul_kernel_descriptor_persistent_bf16(s, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, mi)
    ref = A.float() @ B.float().t()
    assert (C.float() - ref).abs().max() / ref.abs().max() < 1e-2
    # flash-decode split + combine through the AOT entries vs the PyTorch oracle
    from triton_distributed_gh_amd.ops.flash_decode import decode_reference

    batch, Hq, Hkv, D, page, splits, L = 3, 24, 3, 128, 1, 32, 2000
    q = torch.randn(batch, Hq, D, device="cuda").bfloat16()
    kc = torch.randn(batch * L, page, Hkv, D, device="cuda").bfloat16()
    vc = torch.randn(batch * L, page, Hkv, D, device="cuda").bfloat16()
    table = torch.arange(batch * L, device="cuda", dtype=torch.int32).view(batch, L)
    lens = torch.tensor([L, 777, 1], device="cuda", dtype=torch.int32)
    part = torch.empty(batch, Hq, splits, D + 1, device="cuda", dtype=torch.float32)
    out = torch.empty(batch, Hq, D, device="cuda", dtype=torch.bfloat16)
    si = distributed.gqa_fwd_batch_decode_split_kv_bf16_bf16_fp32__triton_algo_info_t(
        **aot_kernels.get_split_kv_algo_info(Hq, Hkv, D, D, page, splits))
    ci = distributed.gqa_fwd_batch_decode_combine_kv_fp32_bf16__triton_algo_info_t(**aot_kernels.get_combine_kv_algo_info(splits, D))
    distributed.gqa_fwd_batch_decode_split_kv_bf16_bf16_fp32(
        s, q.data_ptr(), kc.data_ptr(), vc.data_ptr(), part.data_ptr(), D ** -0.5, table.data_ptr(), lens.data_ptr(), batch,
        q.stride(0), q.stride(1), kc.stride(-3), kc.stride(-2), vc.stride(-3), vc.stride(-2), part.stride(0),
        part.stride(1), part.stride(2), table.stride(0), si)
    distributed.gqa_fwd_batch_decode_combine_kv_fp32_bf16(s, part.data_ptr(), out.data_ptr(), lens.data_ptr(), batch, Hq,
                                                          part.stride(0), part.stride(1), part.stride(2), out.stride(0),
                                                          out.stride(1), ci)
    ref = decode_reference(q, kc, vc, lens, table, D ** -0.5)
    assert (out.float() - ref).abs().max() < 3e-2


@pytest.mark.gpu
@pytest.mark.parametrize("t,dtype", [("fp16", torch.float16), ("bf16", torch.bfloat16)])
def test_aot_16_bit_partials_and_persistent_entry(t, dtype):
    """The reference's ``*_fp16_fp16_fp16`` kernels (flash_decode.py:112-128, :289-305, :364-385, :578-586): split
    results kept in the 16-bit type, merged by the combine, the intra-rank combine (LSE column) and the
    single-launch persistent entry; all against the PyTorch oracle."""
    from triton_distributed_gh_amd.ops.flash_decode import decode_reference

    s = torch.cuda.current_stream().cuda_stream
    batch, Hq, Hkv, D, page, splits, L = 3, 64, 8, 128, 64, 32, 1536
    pages = L // page
    q = torch.randn(batch, Hq, D, device="cuda").to(dtype)
    kc = torch.randn(batch * pages, page, Hkv, D, device="cuda").to(dtype)
    vc = torch.randn(batch * pages, page, Hkv, D, device="cuda").to(dtype)
    table = torch.randperm(batch * pages, device="cuda").int().view(batch, pages)
    lens = torch.tensor([L, 777, 1], device="cuda", dtype=torch.int32)
    ref, ref_lse = decode_reference(q, kc, vc, lens, table, D ** -0.5, return_lse=True)
    tol = 3e-2 if dtype == torch.float16 else 6e-2       # the LSE weights go through 16 bits as well

    part = torch.empty(batch, Hq, splits, D + 1, device="cuda", dtype=dtype)
    out = torch.empty(batch, Hq, D, device="cuda", dtype=dtype)
    info = aot_kernels.get_split_kv_algo_info(Hq, Hkv, D, D, page, splits)
    split = getattr(distributed, f"gqa_fwd_batch_decode_split_kv_{t}_{t}_{t}")
    si = getattr(distributed, f"gqa_fwd_batch_decode_split_kv_{t}_{t}_{t}__triton_algo_info_t")(**info)
    split(s, q.data_ptr(), kc.data_ptr(), vc.data_ptr(), part.data_ptr(), D ** -0.5, table.data_ptr(), lens.data_ptr(), batch,
          q.stride(0), q.stride(1), kc.stride(-3), kc.stride(-2), vc.stride(-3), vc.stride(-2), part.stride(0), part.stride(1),
          part.stride(2), table.stride(0), si)
    cinfo = aot_kernels.get_combine_kv_algo_info(splits, D)
    ci = getattr(distributed, f"gqa_fwd_batch_decode_combine_kv_{t}_{t}__triton_algo_info_t")(**cinfo)
    getattr(distributed, f"gqa_fwd_batch_decode_combine_kv_{t}_{t}")(
        s, part.data_ptr(), out.data_ptr(), lens.data_ptr(), batch, Hq, part.stride(0), part.stride(1), part.stride(2),
        out.stride(0), out.stride(1), ci)
    assert (out.float() - ref).abs().max() < tol

    out_lse = torch.empty(batch, Hq, D + 1, device="cuda", dtype=dtype)
    ii = getattr(distributed, f"intra_rank_gqa_fwd_batch_decode_combine_kv_{t}_{t}__triton_algo_info_t")(**cinfo)
    getattr(distributed, f"intra_rank_gqa_fwd_batch_decode_combine_kv_{t}_{t}")(
        s, part.data_ptr(), out_lse.data_ptr(), lens.data_ptr(), batch, Hq, part.stride(0), part.stride(1), part.stride(2),
        out_lse.stride(0), out_lse.stride(1), ii)
    assert (out_lse[..., :D].float() - ref).abs().max() < tol
    assert (out_lse[..., D].float() - ref_lse).abs().max() < 5e-2

    # persistent entry: one launch, workspace = arrival counters (zero before the first launch, left zero)
    part.zero_()
    final = torch.zeros(batch, Hq, D, device="cuda", dtype=dtype)
    workspace = torch.zeros(batch * Hkv, device="cuda", dtype=torch.int32)
    pinfo = aot_kernels.get_persistent_algo_info(Hq, Hkv, D, D, page, splits)
    pi = getattr(distributed, f"gqa_fwd_batch_decode_split_kv_persistent_{t}_{t}_{t}__triton_algo_info_t")(**pinfo)
    for _ in range(2):
        getattr(distributed, f"gqa_fwd_batch_decode_split_kv_persistent_{t}_{t}_{t}")(
            s, q.data_ptr(), kc.data_ptr(), vc.data_ptr(), part.data_ptr(), final.data_ptr(), D ** -0.5, table.data_ptr(),
            lens.data_ptr(), workspace.data_ptr(), batch, q.stride(0), q.stride(1), kc.stride(-3), kc.stride(-2),
            vc.stride(-3), vc.stride(-2), part.stride(0), part.stride(1), part.stride(2), final.stride(0), final.stride(1),
            table.stride(0), pi)
        assert (final.float() - ref).abs().max() < tol
        assert int(workspace.abs().sum()) == 0
