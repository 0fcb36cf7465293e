#!/bin/bash
# rocprofv3 kernel durations of the solo-rank probe, one world size per run
set -o pipefail
R=$GRAFT_REPO_ROOT
O=gpurun_out/r2/solo_kt
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
for w in 1 2 4 8; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/$O/kt$w -o kt -- python3 $R/scripts/dev/probe_solo_rank.py $w > $R/$O/kt$w.log 2>&1
  rc=$?; grep "^w=" $R/$O/kt$w.log; [ $rc -ne 0 ] && exit $rc
  python3 $R/scripts/rocpd_kernel_stats.py $(find $R/$O/kt$w -name '*results.db' | head -1) --skip-first 8 > $R/$O/kernel_stats_w$w.txt 2>&1
  head -n 7 $R/$O/kernel_stats_w$w.txt | cut -c1-140
  find $R/$O/kt$w -name '*.db' -delete
done
