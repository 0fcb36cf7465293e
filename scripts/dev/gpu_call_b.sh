#!/bin/bash
# GPU call B: tests after the in-kernel counter re-arm + self-test + bench contract, then a capture-only graph dump.
set -o pipefail
mkdir -p gpurun_out/r2
export GPU_MAX_HW_QUEUES=8
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > gpurun_out/r2/gpu_tests_b.log 2>&1
rc=$?
tail -n 25 gpurun_out/r2/gpu_tests_b.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 > gpurun_out/r2/bench_b.log 2>&1 || { tail -n 20 gpurun_out/r2/bench_b.log; exit 1; }
tail -n 1 gpurun_out/r2/bench_b.log
timeout -k 10 200 python -u scripts/dev/dump_capture_graph.py gpurun_out/r2 > gpurun_out/r2/dump_b.log 2>&1
rc=$?
tail -n 30 gpurun_out/r2/dump_b.log
exit $rc
