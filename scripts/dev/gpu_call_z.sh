#!/bin/bash
# decode step against batch size: norm folded into the projections (up to M rows) or not, split-K decode kernel, library for big layers
set -o pipefail
O=gpurun_out/r2/z
mkdir -p $O
run() {  # name, bsz, env...
  local name=$1 B=$2; shift 2
  env "$@" timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz $B > $O/${name}_bs$B.log 2>&1 || { tail -30 $O/${name}_bs$B.log; exit 1; }
  echo "$name bsz $B: $(tail -1 $O/${name}_bs$B.log | cut -c60-175)"
}
for B in 2 4 8 16; do
  run fold $B TDIST_NORM_GEMM_MAX_M=16
  run nofold $B TDIST_NORM_GEMM_MAX_M=0
done
run nofold_splitk 16 TDIST_NORM_GEMM_MAX_M=0 TDIST_DECODE_SPLITK=1
run nofold_lib96 16 TDIST_NORM_GEMM_MAX_M=0 TDIST_DECODE_GEMM_LIB_MB=96
run nofold_lib30 16 TDIST_NORM_GEMM_MAX_M=0 TDIST_DECODE_GEMM_LIB_MB=30
run nofold_splitk 8 TDIST_NORM_GEMM_MAX_M=0 TDIST_DECODE_SPLITK=1
