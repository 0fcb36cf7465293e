#!/bin/bash
# headline bench at one GPU + kernel-trace stats of the same run, smoke(), and the e2e decode bench
set -o pipefail
R=$GRAFT_REPO_ROOT
O=gpurun_out/r2/final
mkdir -p $R/$O
cd $R
timeout -k 10 200 python -u bench.py --gpus 1 --steps 50 --warmup 10 > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
cat $O/bench_n1.json
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || exit 1
tail -n 2 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python3 $R/bench.py --gpus 1 --steps 50 --warmup 10 > $R/$O/prof.log 2>&1 || exit 1
python3 $R/scripts/rocpd_kernel_stats.py $(find $R/$O/prof -name '*results.db' | head -1) > $R/$O/bench_kernel_stats.txt 2>&1
head -n 20 $R/$O/bench_kernel_stats.txt | cut -c1-220
find $R/$O -name '*.db' -delete
