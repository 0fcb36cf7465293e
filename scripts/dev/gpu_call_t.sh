#!/bin/bash
# Engine.serve on one GPU: Qwen3-8B, batch 1 and batch 4, torch backend against triton_dist_AR
set -o pipefail
R=$GRAFT_REPO_ROOT
O=gpurun_out/r2/engine_e2e
mkdir -p $R/$O
cd $R
for B in 1 4; do
  for BK in torch triton_dist_AR; do
    timeout -k 10 250 python -u triton_dist/test/amd/test_e2e_inference.py --model Qwen/Qwen3-8B --bsz $B --prompt_len 128 --gen_len 64 --max_length 256 --backend $BK > $O/b${B}_$BK.log 2>&1 || { tail -8 $O/b${B}_$BK.log; exit 1; }
    echo "bsz=$B $BK: $(grep -i "ms\|tok" $O/b${B}_$BK.log | tail -n 3 | tr '\n' ' ' | cut -c1-300)"
  done
done
