#!/bin/bash
# A/B of two builds of the GPU library on one box for the mega interpreter (Qwen3-8B, batch 1)
set -o pipefail
L=triton_distributed_gh_amd
cp $L/libtdist_hip.so /tmp/new.so && cp $L/libtdist_hip_old.so /tmp/old.so
for V in old new old new; do
  cp /tmp/$V.so $L/libtdist_hip.so
  echo "== $V: $(timeout -k 10 250 python -u benchmarks/bench_qwen3_mega.py --bsz ${BSZ:-1} | tail -n 1 | grep -o '"mega_interpreter_ms": [0-9.]*')" || exit 1
done
cp /tmp/new.so $L/libtdist_hip.so
