#!/bin/bash
# decode step: norm folded up to M rows AND the other long-K projections on the split-K kernel
set -o pipefail
O=gpurun_out/r2/z4
mkdir -p $O
run() {  # name, bsz, env...
  local name=$1 B=$2; shift 2
  env "$@" timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz $B > $O/${name}_bs$B.log 2>&1 || { tail -30 $O/${name}_bs$B.log; exit 1; }
  echo "$name bsz $B: $(tail -1 $O/${name}_bs$B.log | cut -c80-155)"
}
run fold16_splitk2 2 TDIST_NORM_GEMM_MAX_M=16 TDIST_DECODE_SPLITK_MIN_M=2
run fold16_splitk2 3 TDIST_NORM_GEMM_MAX_M=16 TDIST_DECODE_SPLITK_MIN_M=2
run fold1_splitk2 3 TDIST_NORM_GEMM_MAX_M=1 TDIST_DECODE_SPLITK_MIN_M=2
run fold16 3 TDIST_NORM_GEMM_MAX_M=16
run fold16_splitk2 4 TDIST_NORM_GEMM_MAX_M=16 TDIST_DECODE_SPLITK_MIN_M=2
run fold16_splitk2 8 TDIST_NORM_GEMM_MAX_M=16 TDIST_DECODE_SPLITK_MIN_M=2
run fold16 8 TDIST_NORM_GEMM_MAX_M=16
