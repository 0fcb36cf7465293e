#!/bin/bash
# A/B of two builds of the GPU library on one box for the mid-M GEMM with cold weights
set -o pipefail
L=triton_distributed_gh_amd
cp $L/libtdist_hip.so /tmp/new.so && cp $L/libtdist_hip_old.so /tmp/old.so
for V in old new old new; do
  cp /tmp/$V.so $L/libtdist_hip.so
  echo "== $V"; timeout -k 10 200 python -u scripts/dev/probe_mid_cold.py 2>&1 | grep -E "M=|rror" || exit 1
done
cp /tmp/new.so $L/libtdist_hip.so
