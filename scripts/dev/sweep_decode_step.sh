#!/bin/bash
# decode step of Qwen3-8B against batch size and context (HIP graph), current defaults
set -o pipefail
O=gpurun_out/sweep
mkdir -p $O
for B in 1 2 4 8 16; do
  timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz $B > $O/bs$B.log 2>&1 || { tail -30 $O/bs$B.log; exit 1; }
  echo "bsz $B: $(tail -1 $O/bs$B.log | cut -c80-200)"
done
for C in 2048 8192; do
  timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz 1 --ctx $C > $O/ctx$C.log 2>&1 || { tail -30 $O/ctx$C.log; exit 1; }
  echo "ctx $C: $(tail -1 $O/ctx$C.log | cut -c80-200)"
done
