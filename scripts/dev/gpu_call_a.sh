#!/bin/bash
# GPU call A: baseline tests + bench + GEMM A/B, then the straddle-capture reproduction LAST (it may fault).
set -o pipefail
mkdir -p gpurun_out/r2
export GPU_MAX_HW_QUEUES=8
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r2/gpu_tests_a.log 2>&1 \
  && tail -n 3 gpurun_out/r2/gpu_tests_a.log \
  && timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 > gpurun_out/r2/bench_a.log 2>&1 \
  && tail -n 2 gpurun_out/r2/bench_a.log \
  && timeout -k 10 200 python -u scripts/dev/probe_gemm2.py > gpurun_out/r2/gemm_ab_a.log 2>&1 \
  && tail -n 5 gpurun_out/r2/gemm_ab_a.log \
  && timeout -k 10 240 python -u scripts/dev/repro_capture_straddle.py all > gpurun_out/r2/repro_a.log 2>&1
rc=$?
echo "rc=$rc"
tail -n 40 gpurun_out/r2/repro_a.log
exit $rc
