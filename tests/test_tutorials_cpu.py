"""Tutorials and command-line scripts run end to end on the CPU backend with virtual ranks."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CASES = [
    ["tutorials/01-distributed-notify-wait.py", "--virtual-ranks", "3"],
    ["tutorials/02-intra-node-allgather.py", "--virtual-ranks", "2", "--rows", "32", "--cols", "64"],
    ["tutorials/05-intra-node-reduce-scatter.py", "--virtual-ranks", "4", "--M", "64", "--N", "32"],
    ["tutorials/09-AMD-overlapping-allgather-gemm.py", "--virtual-ranks", "2", "--M", "128", "--N", "64", "--K", "64",
     "--iters", "1", "--warmup", "1"],
    ["tutorials/10-AMD-overlapping-gemm-reduce-scatter.py", "--virtual-ranks", "2", "--M", "128", "--N", "64", "--K", "64",
     "--iters", "1", "--warmup", "1"],
    ["triton_dist/test/amd/test_ag_gemm_intra_node.py", "256", "128", "64", "--virtual-ranks", "2", "--check"],
    ["triton_dist/test/amd/test_gemm_rs_intra_node.py", "256", "128", "64", "--virtual-ranks", "2", "--check"],
    ["triton_dist/test/amd/test_barrier_all.py", "--virtual-ranks", "3", "--rounds", "3"],
    ["triton_dist/test/nvidia/test_ag_small_msg.py", "--virtual-ranks", "3", "--check", "--max-nbytes", "4096"],
    ["triton_dist/test/nvidia/test_notify.py", "--virtual-ranks", "3", "--rounds", "2"],
    ["triton_dist/test/nvidia/test_ring_put.py", "--virtual-ranks", "4", "-n", "256"],
    ["triton_dist/test/nvidia/test_multimem_ld_reduce.py", "--virtual-ranks", "3", "-n", "1000"],
    ["triton_dist/test/nvidia/test_nvshmem_api.py", "-n", "64"],
    ["triton_dist/test/nvidia/test_team_split.py"],
    ["triton_dist/test/nvidia/test_simt.py"],
    ["triton_dist/test/nvidia/test_utils.py"],
]


@pytest.mark.parametrize("case", CASES, ids=[c[0].split("/")[-1] for c in CASES])
def test_script_runs(case):
    env = {"CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": "", "PATH": "/usr/bin:/bin:/usr/local/bin:/opt/rocm/bin",
           "RANDOM_PARAMS": "1", "TDIST_NO_HF_CONFIG": "1", "HOME": "/tmp"}
    res = subprocess.run([sys.executable, *case], cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0 and "✅" in res.stdout, (res.stdout + res.stderr)[-2000:]
