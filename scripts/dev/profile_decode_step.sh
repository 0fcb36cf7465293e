#!/bin/bash
# Per-kernel times and the kernel timeline of the graph-replayed decode step (Qwen3-8B shapes, 6 layers).
#   BATCHES="1 4" scripts/dev/profile_decode_step.sh      (default: batch 1 and 4)
# Writes gpurun_out/decode_bs<B>/{run.log, stats.txt, timeline_full.txt}.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for B in ${BATCHES:-1 4}; do
  O=gpurun_out/decode_bs$B
  mkdir -p $R/$O
  timeout -s KILL 280 rocprofv3 --kernel-trace -d $R/$O/prof -o p -- python3 $R/benchmarks/bench_qwen3_mega.py --model ${MODEL:-Qwen/Qwen3-8B} --layers 6 --iters 10 --bsz $B > $R/$O/run.log 2>&1 || { tail -5 $R/$O/run.log; exit 1; }
  tail -n 1 $R/$O/run.log
  DB=$(find $R/$O/prof -name '*results.db' | head -1)
  python3 $R/scripts/rocpd_kernel_stats.py $DB > $R/$O/stats.txt 2>&1
  python3 $R/scripts/rocpd_timeline.py $DB --last 3000 --full > $R/$O/timeline_full.txt 2>&1
  head -n 30 $R/$O/stats.txt | cut -c1-200
  find $R/$O -name '*.db' -delete
done
