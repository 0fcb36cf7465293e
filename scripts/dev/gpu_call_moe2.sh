#!/bin/bash
set -o pipefail
O=gpurun_out/r2/moe_e2e
mkdir -p $O
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_moe_gpu.py -k "decode" tests/test_gemm_gpu.py -k "decode or silu or norm" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_moe_gpu.py tests/test_layers_gpu.py > $O/t2.log 2>&1 || { tail -30 $O/t2.log; exit 1; }
tail -2 $O/t2.log
bash scripts/dev/gpu_call_moe.sh
