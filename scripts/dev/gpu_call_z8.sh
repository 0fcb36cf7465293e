#!/bin/bash
set -o pipefail
O=gpurun_out/r2/z8
mkdir -p $O
timeout -k 10 400 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_attention_gpu.py tests/test_layers_gpu.py tests/test_mega.py tests/test_tp_ll_gpu.py > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
timeout -k 10 700 python -u scripts/dev/probe_decode_splits.py > $O/probe.log 2>&1 || { tail -30 $O/probe.log; exit 1; }
grep -E "B=|len" $O/probe.log
