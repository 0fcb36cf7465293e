#!/bin/bash
# long race hunt: fused TP ops under hostile timing at 2 / 4 / 8 virtual ranks, and the AG-GEMM stress script
set -o pipefail
R=$GRAFT_REPO_ROOT
O=gpurun_out/stress
mkdir -p $R/$O
cd $R
for W in 8; do
  TDIST_DEBUG_RACES=1 timeout -k 10 280 python -u triton_dist/test/stress/stress_race_debug.py --virtual-ranks $W --iters 600 --K 2048 --N 2048 > $O/race_w$W.log 2>&1 || { tail -8 $O/race_w$W.log; exit 1; }
  echo "race-debug w=$W: $(tail -n 1 $O/race_w$W.log | cut -c1-200)"
done
timeout -k 10 280 python -u triton_dist/test/stress/stress_test_ag_gemm.py --virtual-ranks 8 --iters 300 --simulate_straggler > $O/ag_stress.log 2>&1 || { tail -8 $O/ag_stress.log; exit 1; }
echo "ag stress: $(tail -n 1 $O/ag_stress.log | cut -c1-200)"
