#!/usr/bin/env bash
# Sweep the LLM head shapes of the reference's my_lab/benchmark/bench_matmul.sh over batch sizes.
# usage: bash my_lab/benchmark/bench_matmul.sh [results.csv]
set -euo pipefail
cd "$(dirname "$0")"
out=${1:-benchmark_results.csv}
configs=(
  "mistral-7b,4096,32768"
  "llama3-8b,4096,128256"
  "llama3-70b,8192,12800"
  "qwen3-30b,2048,151936"
  "gpt-neo,6144,50251"
  "Gemma-7b,3072,256000"
)
for conf in "${configs[@]}"; do
  IFS=',' read -r model inf outf <<< "$conf"
  for batch in 1 8 16 32 64 128; do
    python matmul.py --batch "$batch" --in_features "$inf" --out_features "$outf" --model_name "$model" --csv "$out"
  done
done
