#!/bin/bash
set -o pipefail
O=gpurun_out/r2/k1
mkdir -p $O
run() {  # name, bsz, env...
  local name=$1 B=$2; shift 2
  env "$@" timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz $B > $O/${name}_bs$B.log 2>&1 || { tail -30 $O/${name}_bs$B.log; exit 1; }
  echo "$name bsz $B: $(tail -1 $O/${name}_bs$B.log | cut -c80-160)"
}
run base 1 X=1
run splitk1 1 TDIST_DECODE_SPLITK_MIN_M=1
run bn32 1 TDIST_SKINNY_BN=32
run base 1 X=1
run nosplitk 4 TDIST_DECODE_SPLITK_MIN_M=17
run base 4 X=1
