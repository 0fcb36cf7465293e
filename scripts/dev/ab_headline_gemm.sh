#!/bin/bash
# A/B of two builds of the GPU library on one box for the headline GEMM: libtdist_hip_old.so against libtdist_hip.so
set -o pipefail
L=triton_distributed_gh_amd
cp $L/libtdist_hip.so /tmp/new.so && cp $L/libtdist_hip_old.so /tmp/old.so
for V in old new old new old new; do
  cp /tmp/$V.so $L/libtdist_hip.so
  echo "== $V: $(timeout -k 10 100 python -u scripts/dev/time_headline_gemm.py 2>&1 | tail -n 1)" || exit 1
done
cp /tmp/new.so $L/libtdist_hip.so
