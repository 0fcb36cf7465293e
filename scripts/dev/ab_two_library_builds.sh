#!/bin/bash
# A/B of two builds of the native library on one box: libtdist_hip_old.so against libtdist_hip.so
set -o pipefail
O=gpurun_out/ab
mkdir -p $O
L=triton_distributed_gh_amd
cp $L/libtdist_hip.so /tmp/new.so && cp $L/libtdist_hip_old.so /tmp/old.so
for V in old new old new; do
  cp /tmp/$V.so $L/libtdist_hip.so
  echo "== $V"
  timeout -k 10 200 python -u scripts/dev/probe_skinny_ab.py 2>&1 | grep -E "MB|rror" || exit 1
done
for V in old new; do
  cp /tmp/$V.so $L/libtdist_hip.so
  for B in 1 4; do
    timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --bsz $B > $O/${V}_bs$B.log 2>&1 || { tail -30 $O/${V}_bs$B.log; exit 1; }
    echo "$V bsz $B: $(tail -1 $O/${V}_bs$B.log | cut -c80-200)"
  done
done
cp /tmp/new.so $L/libtdist_hip.so
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gemm_gpu.py -k "decode or skinny or norm or silu" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
