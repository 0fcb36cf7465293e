#!/bin/bash
# the whole GPU tier, then smoke and the headline bench
set -o pipefail
mkdir -p gpurun_out/r2/full
timeout -k 10 900 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests -m gpu > gpurun_out/r2/full/gpu_tests.log 2>&1 || { tail -40 gpurun_out/r2/full/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r2/full/gpu_tests.log
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2/full/smoke.log 2>&1 || { tail -30 gpurun_out/r2/full/smoke.log; exit 1; }
tail -2 gpurun_out/r2/full/smoke.log
timeout -k 10 300 python -u bench.py --gpus 1 > gpurun_out/r2/full/bench.log 2>&1 || { tail -30 gpurun_out/r2/full/bench.log; exit 1; }
tail -1 gpurun_out/r2/full/bench.log
