#!/bin/bash
# soak of the decode paths: attention / layer / MoE tests three times over, then long generations on both model families
set -o pipefail
O=gpurun_out/soak
mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 400 python -u -m pytest -x -q --timeout 120 --timeout-method thread -p no:randomly tests/test_attention_gpu.py tests/test_layers_gpu.py tests/test_moe_gpu.py tests/test_gemm_gpu.py > $O/t$i.log 2>&1 || { tail -30 $O/t$i.log; exit 1; }
  echo "pass $i: $(tail -1 $O/t$i.log)"
done
timeout -k 10 300 python -u triton_dist/test/amd/test_e2e_inference.py --model Qwen/Qwen3-8B --num_layers 12 --bsz 3 --prompt_len 100 --gen_len 400 --max_length 512 --backend triton_dist_AR > $O/dense_long.log 2>&1 || { tail -12 $O/dense_long.log; exit 1; }
echo "dense: $(grep generated $O/dense_long.log | tail -1)"
timeout -k 10 300 python -u triton_dist/test/amd/test_e2e_inference.py --model Qwen/Qwen3-30B-A3B --num_layers 6 --bsz 3 --prompt_len 100 --gen_len 400 --max_length 512 --backend triton_dist > $O/moe_long.log 2>&1 || { tail -12 $O/moe_long.log; exit 1; }
echo "moe: $(grep generated $O/moe_long.log | tail -1)"
