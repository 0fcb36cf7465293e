#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
export GPU_MAX_HW_QUEUES=24
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/gpu_tests_h.log 2>&1
rc=$?; tail -n 5 gpurun_out/r2/gpu_tests_h.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u scripts/dev/trace_solo_rank.py 1,2,8 > gpurun_out/r2/trace_solo_h.log 2>&1
rc=$?; cat gpurun_out/r2/trace_solo_h.log; [ $rc -ne 0 ] && exit $rc
O=gpurun_out/r2/solo_kt
R=$GRAFT_REPO_ROOT
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python3 $R/scripts/dev/probe_solo_rank.py 1,2,4,8 > $R/$O/kt.log 2>&1
rc=$?; grep "^w=" $R/$O/kt.log; [ $rc -ne 0 ] && exit $rc
python3 $R/scripts/rocpd_kernel_stats.py $(find $R/$O/kt -name '*results.db' | head -1) --skip-first 8 > $R/$O/kernel_stats.txt 2>&1
head -n 14 $R/$O/kernel_stats.txt | cut -c1-150
find $R/$O -name '*.db' -delete
