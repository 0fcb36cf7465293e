#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
export GPU_MAX_HW_QUEUES=24
timeout -k 10 300 python -u scripts/dev/probe_gemm_kscan.py > gpurun_out/r2/gemm_kscan_1.log 2>&1 || { tail -n 20 gpurun_out/r2/gemm_kscan_1.log; exit 1; }
cat gpurun_out/r2/gemm_kscan_1.log
timeout -k 10 200 python -u -m pytest tests/test_user_kernels.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r2/user_kernels.log 2>&1
rc=$?; tail -n 15 gpurun_out/r2/user_kernels.log; [ $rc -ne 0 ] && exit $rc
TDIST_LOG=0 timeout -k 10 150 python -u tutorials/01-distributed-notify-wait.py --virtual-ranks 4 --heap-mb 64 > gpurun_out/r2/tut01.log 2>&1
rc=$?; tail -n 8 gpurun_out/r2/tut01.log; exit $rc
