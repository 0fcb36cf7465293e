#!/bin/bash
set -o pipefail
O=gpurun_out/r2/moe_e2e
mkdir -p $O
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_moe_gpu.py -k "decode or router" > $O/t6.log 2>&1 || { tail -30 $O/t6.log; exit 1; }
tail -2 $O/t6.log
bash scripts/dev/gpu_call_moe.sh && bash scripts/dev/gpu_call_moe3.sh > $O/trace.log 2>&1 && grep -E "moe_decode_down|skinny_kernel<true, 2, 4, true, false>" $O/trace.log | tail -3
