#!/bin/bash
# kernel timeline of the graph-replayed decode step (native layers), Qwen3-8B shapes, 6 layers
set -o pipefail
R=$GRAFT_REPO_ROOT
O=gpurun_out/r2/decode_tl
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace -d $R/$O/prof -o p -- python3 $R/benchmarks/bench_qwen3_mega.py --model Qwen/Qwen3-8B --layers 6 --iters 10 > $R/$O/run.log 2>&1 || { tail -5 $R/$O/run.log; exit 1; }
tail -n 3 $R/$O/run.log
DB=$(find $R/$O/prof -name '*results.db' | head -1)
python3 $R/scripts/rocpd_kernel_stats.py $DB > $R/$O/stats.txt 2>&1
python3 $R/scripts/rocpd_timeline.py $DB --last 3000 --full > $R/$O/timeline_full.txt 2>&1
head -n 40 $R/$O/stats.txt | cut -c1-200
find $R/$O -name '*.db' -delete
