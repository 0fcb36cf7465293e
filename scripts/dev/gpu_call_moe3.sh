#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=gpurun_out/r2/moe_trace
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace -d $R/$O/prof -o p -- python3 $R/scripts/dev/run_moe_decode_once.py > $R/$O/run.log 2>&1 || { tail -15 $R/$O/run.log; exit 1; }
DB=$(find $R/$O/prof -name '*results.db' | head -1)
python3 $R/scripts/rocpd_timeline.py $DB --last 150 --full > $R/$O/timeline.txt 2>&1
find $R/$O -name '*.db' -delete
tail -45 $R/$O/timeline.txt | cut -c1-190
