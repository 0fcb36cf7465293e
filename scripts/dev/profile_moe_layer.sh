#!/bin/bash
# Kernel times (rocprofv3 kernel trace) of one Qwen3-30B-A3B MoE layer at a given token count.
# usage: profile_moe_layer.sh <tokens> [SHAPE as E,topk,H,I]
set -e
tok=${1:-4096}
export SHAPE=${2:-128,8,2048,768} CTX_M=8192 NO_TORCH=1 TOKENS=$tok
out=gpurun_out/r3/moe_layer_prof_$tok
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out -o run -- python3 $GRAFT_REPO_ROOT/scripts/dev/probe_moe_layer_tokens.py > $GRAFT_REPO_ROOT/$out/probe.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_kernel_stats.py $out/run_results.db > $out/kernel_stats.txt 2>&1 || ls -R $out | head -30
head -n 24 $out/kernel_stats.txt | cut -c1-200
