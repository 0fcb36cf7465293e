#!/bin/bash
# fused decode step (prep inside the attention kernel): numerics, then layer / model tests, then the decode bench
set -o pipefail
mkdir -p gpurun_out/u
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_attention_gpu.py -k "decode_step_from_qkv or qkv_prep" > gpurun_out/u/t1.log 2>&1 || { tail -30 gpurun_out/u/t1.log; exit 1; }
tail -3 gpurun_out/u/t1.log
timeout -k 10 400 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_layers_gpu.py tests/test_attention_gpu.py > gpurun_out/u/t2.log 2>&1 || { tail -30 gpurun_out/u/t2.log; exit 1; }
tail -3 gpurun_out/u/t2.log
timeout -k 10 300 python -u benchmarks/bench_qwen3_mega.py > gpurun_out/u/bench_fused.log 2>&1 || { tail -30 gpurun_out/u/bench_fused.log; exit 1; }
tail -8 gpurun_out/u/bench_fused.log
TDIST_DECODE_FUSED_PREP=0 timeout -k 10 300 python -u benchmarks/bench_qwen3_mega.py > gpurun_out/u/bench_two.log 2>&1 || { tail -30 gpurun_out/u/bench_two.log; exit 1; }
tail -8 gpurun_out/u/bench_two.log
