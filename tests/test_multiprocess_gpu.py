"""One process per rank over the hipIpc symmetric heap (two ranks sharing the one GPU): the launch
path of the benchmark and of every command-line script, as opposed to the in-process virtual ranks
the other GPU tests use.  Scenario: the reference's torchrun-launched tests (scripts/launch_amd.sh)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _torchrun(nproc, script, *args, port=29571, timeout=240):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TDIST_LOG="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script), *args]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_two_processes_run_the_fused_path():
    res = _torchrun(2, ROOT / "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-baseline")
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["timed_path"] == "fused HIP kernels" and out["note"] == "", out


def test_cli_script_two_processes_check():
    res = _torchrun(2, ROOT / "triton_dist/test/amd/test_gemm_rs_intra_node.py", "1024", "1024", "1024", "--check",
                    "--heap-mb", "3", port=29572)      # 3 MiB segments: the scatter buffers force the heap to grow
    assert res.returncode == 0 and "check passed" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]
