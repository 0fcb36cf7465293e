"""How the tile primitives are lowered to gfx950 instructions (no GPU needed: hipcc cross-compiles
and the test reads the assembly).  Counterpart of the reference's MLIR lit test of the
``distributed.wait`` lowering (unittest/lower_wait.mlir:58) and of
test/nvidia/test_distributed_wait.py --case lower."""
import shutil

import pytest

from triton_distributed_gh_amd.tools import isa

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")


def test_wait_and_notify_lowering():
    report = isa.lowering_report()
    failed = [fact for fact, ok in report.items() if not ok]
    assert not failed, failed


def test_write_through_tile_stores_carry_cache_policy_bits():
    # the GEMM+RS epilogue publishes tiles with system-scope write-through stores and no L2 write-back
    body = isa.kernel_isa("gemm_rs.hip", "gemm_rs_kernelILi1")
    assert "global_store_dwordx4" in body and "sc0 sc1" in body
    stores = [ln for ln in body.splitlines() if ln.startswith("global_store_dwordx4")]
    assert any(ln.endswith("sc0 sc1") for ln in stores)
    assert "buffer_wbl2" not in body, "the fused GEMM+RS kernel must not execute a release fence per tile"


def test_gemm_main_loop_uses_mfma_and_direct_to_lds_loads():
    body = isa.kernel_isa("gemm.hip", "gemm_nt_256_kernelILb1ELi2E")
    assert body.count("v_mfma_f32_16x16x32_bf16") >= 64           # 2 phases x 32 MFMAs per wave and K-step
    assert "global_load_lds_dwordx4" in body
    assert "ds_read_b128" in body or "ds_read2_b64" in body
    assert "scratch_" not in body, "no register spills in the main loop"
