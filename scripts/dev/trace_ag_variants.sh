#!/bin/bash
# Per-workgroup trace of the AG+GEMM consumer at one world size under the plan knobs.
# usage: trace_ag_variants.sh <world> "<producer> <split_k|0> <qn|0> <tile|0> [acquire mode]" ...   (0 = the plan's own choice)
set -e
export GPU_MAX_HW_QUEUES=24
w=${1:-8}; shift
mkdir -p gpurun_out/r3/ag_variants
for v in "$@"; do
  set -- $v
  name=prod_$1.split_$2.qn_$3.tile_$4.acq_${5:-0}
  export TDIST_AG_ACQUIRE=${5:-0}
  env PROBE_PRODUCER=$1 $( [ $2 != 0 ] && echo TDIST_AG_SPLIT_K=$2 ) $( [ $3 != 0 ] && echo TDIST_AG_QN=$3 ) $( [ $4 != 0 ] && echo TDIST_AG_TILE=$4 ) \
    timeout -k 10 120 python -u scripts/dev/trace_solo_rank.py $w > gpurun_out/r3/ag_variants/w$w.$name.log 2>&1
  echo "== $name"; grep -A7 "ag_gemm" gpurun_out/r3/ag_variants/w$w.$name.log | grep -v gemm_rs
done
