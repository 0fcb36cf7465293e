#!/bin/bash
# Qwen3-8B decode step (HIP graph) against the row count up to which the RMSNorm rides inside the projection launch
set -o pipefail
O=gpurun_out/r3/norm_thr
mkdir -p $O
for B in 4 8 16; do
  for T in 16 8 4 2; do
    TDIST_NORM_GEMM_MAX_M=$T timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz $B > $O/bs$B.t$T.log 2>&1 || { tail -30 $O/bs$B.t$T.log; exit 1; }
    echo "bsz $B max_m $T: $(tail -1 $O/bs$B.t$T.log | cut -c1-260)"
  done
done
