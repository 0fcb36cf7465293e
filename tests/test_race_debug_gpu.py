"""The fused TP ops under the race-hunting debug mode (TDIST_DEBUG_RACES=1): random producer delays, entry
checks of the flag words, NaN-poisoned parity buffers, a random late rank, 200 fresh-input iterations with an
element-wise check.  The mode is read from the environment when the library loads, so the run is a child process
(reference scenarios: python/triton_dist/test/amd/test_distributed-notify-wait.py:39-212,
test/stress/stress_test_ag_gemm.py)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,iters", [(4, 200), (2, 100)])
def test_fused_tp_ops_survive_hostile_timing(w, iters):
    env = dict(os.environ, TDIST_DEBUG_RACES="1", TDIST_LOG="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-u", str(ROOT / "triton_dist/test/stress/stress_race_debug.py"),
                          "--virtual-ranks", str(w), "--iters", str(iters), "--heap-mb", "256"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "race debug: on" in res.stdout and "race hunt passed" in res.stdout, res.stdout[-2000:]
