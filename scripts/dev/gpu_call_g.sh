#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
export GPU_MAX_HW_QUEUES=24
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/gpu_tests_g.log 2>&1
rc=$?; tail -n 12 gpurun_out/r2/gpu_tests_g.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u scripts/dev/trace_solo_rank.py 1,2,8 > gpurun_out/r2/trace_solo_g.log 2>&1
rc=$?; cat gpurun_out/r2/trace_solo_g.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u scripts/dev/probe_solo_rank.py 1,2,4,8 > gpurun_out/r2/solo_rank_g.log 2>&1
rc=$?; tail -n 30 gpurun_out/r2/solo_rank_g.log; exit $rc
