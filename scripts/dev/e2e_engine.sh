#!/bin/bash
# Engine.serve on one GPU, torch backend against this package's: `e2e_engine.sh dense` (Qwen3-8B, triton_dist_AR)
# or `e2e_engine.sh moe` (Qwen3-30B-A3B shapes, 8 layers, triton_dist); batch 1 and 4 (BATCHES="1 4 16 64" for others).
set -o pipefail
KIND=${1:-dense}
O=gpurun_out/engine_e2e_$KIND
mkdir -p $O
if [ "$KIND" = moe ]; then
  ARGS="--model Qwen/Qwen3-30B-A3B --num_layers 8 --prompt_len 128 --gen_len 32 --max_length 256"; BACKENDS="torch triton_dist"
else
  ARGS="--model Qwen/Qwen3-8B --prompt_len 128 --gen_len 64 --max_length 256"; BACKENDS="torch triton_dist_AR"
fi
for B in ${BATCHES:-1 4}; do
  for BK in $BACKENDS; do
    timeout -k 10 400 python -u triton_dist/test/amd/test_e2e_inference.py $ARGS --bsz $B --backend $BK > $O/b${B}_$BK.log 2>&1 || { tail -12 $O/b${B}_$BK.log; exit 1; }
    echo "bsz=$B $BK: $(grep -i "generated\|ms\|tok" $O/b${B}_$BK.log | tail -n 3 | tr '\n' ' ' | cut -c1-300)"
  done
done
