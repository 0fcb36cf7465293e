#!/bin/bash
# outputs kept under profiles/r2: decode attention against split count (both merge forms), decode GEMMs of a layer
set -o pipefail
O=gpurun_out/r2/save
mkdir -p $O
timeout -k 10 700 python -u scripts/dev/probe_decode_splits.py > $O/decode_attention_splits.txt 2>&1 || { tail -30 $O/decode_attention_splits.txt; exit 1; }
timeout -k 10 300 python -u scripts/dev/probe_skinny_ab.py > $O/decode_gemm_layer_shapes.txt 2>&1 || { tail -30 $O/decode_gemm_layer_shapes.txt; exit 1; }
tail -5 $O/decode_gemm_layer_shapes.txt
