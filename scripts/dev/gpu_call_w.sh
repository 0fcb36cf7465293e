#!/bin/bash
# per-kernel times of the graph-replayed decode step at batch 1 and batch 4 (Qwen3-8B shapes, 6 layers)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for B in 1 4; do
  O=gpurun_out/r2/decode_bs$B
  mkdir -p $R/$O
  timeout -s KILL 280 rocprofv3 --kernel-trace -d $R/$O/prof -o p -- python3 $R/benchmarks/bench_qwen3_mega.py --model Qwen/Qwen3-8B --layers 6 --iters 10 --bsz $B > $R/$O/run.log 2>&1 || { tail -5 $R/$O/run.log; exit 1; }
  tail -n 1 $R/$O/run.log
  DB=$(find $R/$O/prof -name '*results.db' | head -1)
  python3 $R/scripts/rocpd_timeline.py $DB --last 400 --full > $R/$O/timeline_full.txt 2>&1
  find $R/$O -name '*.db' -delete
done
