#!/bin/bash
set -o pipefail
O=gpurun_out/r2/moe_e2e
mkdir -p $O
timeout -k 10 400 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_layers_gpu.py tests/test_moe_gpu.py tests/test_multiprocess_gpu.py > $O/t3.log 2>&1 || { tail -30 $O/t3.log; exit 1; }
tail -2 $O/t3.log
bash scripts/dev/gpu_call_moe.sh
timeout -k 10 250 python -u triton_dist/test/amd/test_e2e_inference.py --model Qwen/Qwen3-8B --bsz 1 --prompt_len 128 --gen_len 64 --max_length 256 --backend triton_dist > $O/dense_td.log 2>&1 || { tail -8 $O/dense_td.log; exit 1; }
echo "dense triton_dist bsz 1: $(grep generated $O/dense_td.log | tail -1)"
