#!/bin/bash
set -o pipefail
O=gpurun_out/r2/z9
mkdir -p $O
timeout -k 10 400 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_attention_gpu.py tests/test_layers_gpu.py tests/test_mega.py tests/test_tp_ll_gpu.py > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
for B in 1 2 4 8 16; do
  timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz $B > $O/bs$B.log 2>&1 || { tail -30 $O/bs$B.log; exit 1; }
  echo "bsz $B: $(tail -1 $O/bs$B.log | cut -c80-200)"
done
TDIST_DECODE_ONE_LAUNCH=1 timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz 1 > $O/one_bs1.log 2>&1 || { tail -30 $O/one_bs1.log; exit 1; }
echo "one-launch bsz 1: $(tail -1 $O/one_bs1.log | cut -c80-200)"
for C in 2048 8192; do
  timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz 1 --ctx $C > $O/ctx$C.log 2>&1 || { tail -30 $O/ctx$C.log; exit 1; }
  echo "ctx $C: $(tail -1 $O/ctx$C.log | cut -c80-200)"
  TDIST_DECODE_ONE_LAUNCH=1 timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz 1 --ctx $C > $O/one_ctx$C.log 2>&1 || { tail -30 $O/one_ctx$C.log; exit 1; }
  echo "one-launch ctx $C: $(tail -1 $O/one_ctx$C.log | cut -c80-200)"
done
