#!/bin/bash
# norm pre-pass with grouped row loads: numerics, then the decode step at batch 1 / 4 / 16 and a KV-split sweep at batch 4
set -o pipefail
O=gpurun_out/r2/x
mkdir -p $O
timeout -k 10 400 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gemm_gpu.py -k "norm" tests/test_layers_gpu.py > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
for B in 1 4 16; do
  timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz $B > $O/bs$B.log 2>&1 || { tail -30 $O/bs$B.log; exit 1; }
  echo "bsz $B: $(tail -1 $O/bs$B.log | cut -c1-230)"
done
for S in 2 4 6; do
  TDIST_DECODE_KV_SPLITS=$S timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz 4 > $O/bs4_s$S.log 2>&1 || { tail -30 $O/bs4_s$S.log; exit 1; }
  echo "bsz 4 splits $S: $(tail -1 $O/bs4_s$S.log | cut -c1-230)"
done
