#!/bin/bash
# Fused TP kernels, one rank of a virtual cluster alone on the GPU: in-kernel producer, then the copy-engine producer
# (the consumer kernel then has every CU, which isolates the compute side of AG+GEMM from the push bandwidth of one GPU).
set -e
export GPU_MAX_HW_QUEUES=24
mkdir -p gpurun_out/r3
timeout -k 10 240 python -u scripts/dev/probe_solo_rank.py 2,4,8 > gpurun_out/r3/solo_kernel_producer.log 2>&1
PROBE_PRODUCER=copy_engine timeout -k 10 240 python -u scripts/dev/probe_solo_rank.py 2,4,8 > gpurun_out/r3/solo_copy_engine.log 2>&1
