#!/bin/bash
# decode step against the number of KV splits at batch 4 / 8 / 16 (ctx 512)
set -o pipefail
O=gpurun_out/r2/z5
mkdir -p $O
run() {  # name, bsz, env...
  local name=$1 B=$2; shift 2
  env "$@" timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz $B > $O/${name}_bs$B.log 2>&1 || { tail -30 $O/${name}_bs$B.log; exit 1; }
  echo "$name bsz $B: $(tail -1 $O/${name}_bs$B.log | cut -c80-155)"
}
for S in 4 6 8 9; do run splits$S 4 TDIST_DECODE_KV_SPLITS=$S; done
for S in 2 3 4 8; do run splits$S 8 TDIST_DECODE_KV_SPLITS=$S; done
for S in 1 2 4; do run splits$S 16 TDIST_DECODE_KV_SPLITS=$S; done
