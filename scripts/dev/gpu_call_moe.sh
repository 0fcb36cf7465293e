#!/bin/bash
# Engine.serve on one GPU: Qwen3-30B-A3B shapes (8 layers), torch backend against triton_dist
set -o pipefail
O=gpurun_out/r2/moe_e2e
mkdir -p $O
for B in 1 4; do
  for BK in torch triton_dist; do
    timeout -k 10 400 python -u triton_dist/test/amd/test_e2e_inference.py --model Qwen/Qwen3-30B-A3B --num_layers 8 --bsz $B --prompt_len 128 --gen_len 32 --max_length 256 --backend $BK > $O/b${B}_$BK.log 2>&1 || { tail -12 $O/b${B}_$BK.log; exit 1; }
    echo "bsz=$B $BK: $(grep -i "generated" $O/b${B}_$BK.log | tail -n 1 | cut -c1-200)"
  done
done
