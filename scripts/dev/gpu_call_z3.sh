#!/bin/bash
# decode step against batch size: rows folded into the norm GEMM, rows sent to the split-K kernel
set -o pipefail
O=gpurun_out/r2/z3
mkdir -p $O
run() {  # name, bsz, env...
  local name=$1 B=$2; shift 2
  env "$@" timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz $B > $O/${name}_bs$B.log 2>&1 || { tail -30 $O/${name}_bs$B.log; exit 1; }
  echo "$name bsz $B: $(tail -1 $O/${name}_bs$B.log | cut -c80-155)"
}
run fold16 1 TDIST_NORM_GEMM_MAX_M=16
for B in 2 4; do
  run fold16 $B TDIST_NORM_GEMM_MAX_M=16
  run fold1 $B TDIST_NORM_GEMM_MAX_M=1
  run fold1_splitk2 $B TDIST_NORM_GEMM_MAX_M=1 TDIST_DECODE_SPLITK_MIN_M=2
done
run fold1_splitk2 8 TDIST_NORM_GEMM_MAX_M=1 TDIST_DECODE_SPLITK_MIN_M=2
run fold1_splitk2 16 TDIST_NORM_GEMM_MAX_M=1 TDIST_DECODE_SPLITK_MIN_M=2
run fold16_splitk1 1 TDIST_NORM_GEMM_MAX_M=16 TDIST_DECODE_SPLITK_MIN_M=1
