#!/bin/bash
set -o pipefail
O=gpurun_out/r2/y
mkdir -p $O
for P in 0 1 0 1; do
  echo "== prepass mode $P"
  TDIST_NORM_PREPASS=$P timeout -k 10 120 python -u scripts/dev/probe_norm_gemm.py 2>&1 | grep "us" || exit 1
done
for P in 0 1; do
  for B in 1 4; do
    TDIST_NORM_PREPASS=$P timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz $B > $O/p${P}_bs$B.log 2>&1 || { tail -30 $O/p${P}_bs$B.log; exit 1; }
    echo "prepass $P bsz $B: $(tail -1 $O/p${P}_bs$B.log | cut -c60-200)"
  done
done
