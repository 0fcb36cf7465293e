#!/bin/bash
# decode step against the number of KV splits of the attention kernel
set -o pipefail
mkdir -p gpurun_out/v
for s in 1 2 4 8 16; do
  TDIST_DECODE_KV_SPLITS=$s timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py > gpurun_out/v/splits_$s.log 2>&1 || { tail -30 gpurun_out/v/splits_$s.log; exit 1; }
  echo "splits $s: $(tail -1 gpurun_out/v/splits_$s.log)"
done
