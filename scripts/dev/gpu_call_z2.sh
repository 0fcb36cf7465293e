#!/bin/bash
# decode step against batch size: which rows go to the split-K kernel, norm folded at one row only
set -o pipefail
O=gpurun_out/r2/z2
mkdir -p $O
run() {  # name, bsz, env...
  local name=$1 B=$2; shift 2
  env "$@" timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz $B > $O/${name}_bs$B.log 2>&1 || { tail -30 $O/${name}_bs$B.log; exit 1; }
  echo "$name bsz $B: $(tail -1 $O/${name}_bs$B.log | cut -c80-155)"
}
for B in 1 2 3 4 6; do
  run fold1 $B TDIST_NORM_GEMM_MAX_M=1
  run fold1_splitk $B TDIST_NORM_GEMM_MAX_M=1 TDIST_DECODE_SPLITK_MIN_M=2
done
run fold2 2 TDIST_NORM_GEMM_MAX_M=2
run fold1_splitk 12 TDIST_NORM_GEMM_MAX_M=1 TDIST_DECODE_SPLITK_MIN_M=2
run fold1_splitk 16 TDIST_NORM_GEMM_MAX_M=1 TDIST_DECODE_SPLITK_MIN_M=2
