#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
export GPU_MAX_HW_QUEUES=24
timeout -k 10 120 scripts/dev/bin/mfma_filler_bench > gpurun_out/r2/mfma_filler_bench.log 2>&1 || { tail -n 20 gpurun_out/r2/mfma_filler_bench.log; exit 1; }
cat gpurun_out/r2/mfma_filler_bench.log
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > gpurun_out/r2/gpu_tests_d.log 2>&1
rc=$?; tail -n 12 gpurun_out/r2/gpu_tests_d.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u scripts/dev/probe_solo_rank.py 1,2 > gpurun_out/r2/solo_rank_d.log 2>&1
rc=$?; tail -n 30 gpurun_out/r2/solo_rank_d.log; exit $rc
