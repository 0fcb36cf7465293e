#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=gpurun_out/r2/normfuse
mkdir -p $R/$O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gemm_gpu.py tests/test_layers_gpu.py -x -q -k "decode or fused or layer or tp_" --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -n 3 $O/tests.log
timeout -k 10 200 python -u benchmarks/bench_qwen3_mega.py --model Qwen/Qwen3-8B --iters 30 > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
grep -o '"native_layers_graph_ms": [0-9.]*\|"torch_graph_ms": [0-9.]*\|"mega[a-z_]*": [0-9.]*' $O/bench.log
