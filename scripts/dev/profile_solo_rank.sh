#!/bin/bash
# Kernel times (rocprofv3 kernel trace) of the fused TP kernels with one rank of a virtual cluster alone on the GPU:
# the host-side timing of probe_solo_rank.py includes the Python launch path, which is longer than a 60 us kernel.
# usage: profile_solo_rank.sh <producer: kernel|none|copy_engine> <world sizes, e.g. 4,8>
set -e
export GPU_MAX_HW_QUEUES=24
prod=${1:-none}; ws=${2:-8}
out=gpurun_out/r3/solo_prof_${prod}_${ws//,/_}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export PROBE_PRODUCER=$prod
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out -o run -- python3 $GRAFT_REPO_ROOT/scripts/dev/probe_solo_rank.py $ws > $GRAFT_REPO_ROOT/$out/probe.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_kernel_stats.py $out/run_results.db > $out/kernel_stats.txt 2>&1 || ls -R $out | head -30
head -n 30 $out/kernel_stats.txt
