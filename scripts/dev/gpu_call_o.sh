#!/bin/bash
# decode step (native layers + graph) against the KV split count of the attention kernel, Qwen3-8B, bs=1, ctx=512
set -o pipefail
R=$GRAFT_REPO_ROOT
O=gpurun_out/r2/decode_splits
mkdir -p $R/$O
cd $R
for S in 8 12 16; do
  TDIST_DECODE_KV_SPLITS=$S timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --model Qwen/Qwen3-8B --iters 30 > $O/s$S.log 2>&1 || { tail -5 $O/s$S.log; exit 1; }
  echo "splits=$S $(grep -o '"native_layers_graph_ms": [0-9.]*' $O/s$S.log) $(grep -o '"torch_graph_ms": [0-9.]*' $O/s$S.log)"
done
