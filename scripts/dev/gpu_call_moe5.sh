#!/bin/bash
set -o pipefail
O=gpurun_out/r2/moe_e2e
mkdir -p $O
timeout -k 10 400 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_layers_gpu.py -k "moe" > $O/t4.log 2>&1 || { tail -40 $O/t4.log; exit 1; }
tail -2 $O/t4.log
timeout -k 10 400 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_layers_gpu.py tests/test_moe_gpu.py > $O/t5.log 2>&1 || { tail -40 $O/t5.log; exit 1; }
tail -2 $O/t5.log
