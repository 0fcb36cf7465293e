#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=gpurun_out/r2/moe
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=24
ITERS=12 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python3 $R/scripts/dev/probe_moe_solo.py > $R/$O/kt.log 2>&1
rc=$?; tail -n 3 $R/$O/kt.log; [ $rc -ne 0 ] && exit $rc
python3 $R/scripts/rocpd_kernel_stats.py $(find $R/$O/kt -name '*results.db' | head -1) --skip-first 2 > $R/$O/kernel_stats.txt 2>&1
head -n 16 $R/$O/kernel_stats.txt | cut -c1-160
find $R/$O -name '*.db' -delete
