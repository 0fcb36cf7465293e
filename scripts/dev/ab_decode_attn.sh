#!/bin/bash
# A/B of two builds of the GPU library on one box for decode attention at long KV (cold K / V)
set -o pipefail
L=triton_distributed_gh_amd
cp $L/libtdist_hip.so /tmp/new.so && cp $L/libtdist_hip_old.so /tmp/old.so
for V in old new old new; do
  cp /tmp/$V.so $L/libtdist_hip.so
  echo "== $V"; ROWS=${ROWS:-1x8192,1x32768,1x131072,16x4096} timeout -k 10 250 python -u scripts/dev/probe_decode_splits.py 2>&1 | grep -E "^B=" | sed 's/   | two launches.*//' | cut -c1-220 || exit 1
done
cp /tmp/new.so $L/libtdist_hip.so
