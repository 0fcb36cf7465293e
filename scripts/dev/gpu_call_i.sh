#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
export GPU_MAX_HW_QUEUES=24
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/gpu_tests_i.log 2>&1
rc=$?; tail -n 5 gpurun_out/r2/gpu_tests_i.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u scripts/dev/trace_solo_rank.py 2,4,8 > gpurun_out/r2/trace_solo_i.log 2>&1
rc=$?; grep -v "plain\|amdgpu.ids" gpurun_out/r2/trace_solo_i.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u scripts/dev/probe_solo_rank.py 1,2,4,8 > gpurun_out/r2/solo_rank_i.log 2>&1
rc=$?; grep "^w=" gpurun_out/r2/solo_rank_i.log; [ $rc -ne 0 ] && exit $rc
for s in 1 2 3 4; do
  TDIST_AG_SPLIT_K=$s timeout -k 10 100 python -u scripts/dev/probe_solo_rank.py 4,8 > gpurun_out/r2/solo_rank_i_s$s.log 2>&1
  rc=$?; echo "split_k=$s"; grep "^w=" gpurun_out/r2/solo_rank_i_s$s.log; [ $rc -ne 0 ] && exit $rc
done
exit 0
