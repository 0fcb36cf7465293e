#!/bin/bash
set -o pipefail
O=gpurun_out/r2/fin
mkdir -p $O
timeout -k 10 400 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_attention_gpu.py tests/test_layers_gpu.py tests/test_mega.py tests/test_tp_ll_gpu.py > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -1 $O/t.log
for C in 512 2048 4096 8192; do
  timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz 1 --ctx $C > $O/ctx$C.log 2>&1 || { tail -30 $O/ctx$C.log; exit 1; }
  echo "ctx $C: $(tail -1 $O/ctx$C.log | cut -c80-170)"
done
