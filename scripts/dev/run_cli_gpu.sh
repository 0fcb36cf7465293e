#!/bin/bash
# Runs every command-line test script once on the GPU (virtual ranks) and records pass / fail.
# Stops at the first time-out, abort or crash (nothing more is started on the GPU after one).
export TDIST_LOG=0 TDIST_NO_HF_CONFIG=1 RANDOM_PARAMS=1
out=${1:-gpurun_out/cli.log}
: > "$out"
run() {
  echo "== $*" >> "$out"
  timeout -k 10 150 python -u "$@" >> "$out" 2>&1
  rc=$?
  echo "-- rc=$rc" >> "$out"
  case $rc in 124|137|134|139) echo "STOP after rc=$rc: $*" >> "$out"; tail -n 30 "$out"; exit 1;; esac
}
N=triton_dist/test/nvidia
A=triton_dist/test/amd
V="--virtual-ranks 4 --heap-mb 256"
run $N/test_allreduce.py $V --check --max-nbytes 1048576
run $N/test_allreduce.py $V --max-nbytes 1048576 --iters 5
run $N/test_gemm_ar.py 256 2048 1024 $V --check
run $N/test_gemm_ar.py 128 2048 2048 $V --low-latency --iters 5
run $N/test_fast_allgather.py $V --check
run $N/test_all_to_all.py -M 64 -N 2048 -G 4 --topk 4 $V --check
run $N/test_all_to_all.py -M 64 -N 2048 -G 4 --topk 4 $V --check --with-scale
run $N/test_ep_a2a.py -M 128 -N 1024 --topk 4 -G 2 $V --check
run $N/test_ag_moe.py --M 1024 --N 1024 --K 512 --G 8 --topk 2 $V --check
run $N/test_moe_reduce_rs.py 512 1024 1024 8 2 $V --check
run $N/test_decode_attn.py --check
run $N/test_decode_attn.py --iters 20
run $N/test_sp_decode_attn.py $V --check --kv-len 1500
run $N/test_sp_ag_attention_intra_node.py $V --check --seqlens 512 1024
run $N/test_reduce_scatter.py 1024 512 $V --check
run $N/test_all_gather.py $V --check --nbytes 262144
run $N/test_common_ops.py $V --rounds 20
run $N/test_moe_utils.py
run $N/test_compile_aot.py
run $N/test_pp.py --virtual-ranks 2 --heap-mb 256 --check
run $N/test_tp_attn.py --model tdist/tiny-dense --bsz 8 --seq_len 32 $V --check
run $N/test_tp_moe.py --M 256 --model tdist/tiny-moe $V --check
run $N/test_ag_small_msg.py $V --check
run $N/test_ag_small_msg.py $V --iters 5
run $N/test_notify.py $V
run $N/test_ring_put.py $V
run $N/test_multimem_ld_reduce.py $V
run $N/test_nvshmem_api.py
run $N/test_team_split.py
run $N/test_simt.py
run $N/test_utils.py
run $N/test_language_extra.py
run $N/test_distributed_wait.py --case lower
run $N/test_distributed_wait.py --case correctness
run $N/test_distributed_wait.py --case perf
run $N/test_perf_ring_reduce_tma.py
run $N/test_ep_moe_inference.py -M 64 -N 256 -I 128 -G 2 --topk 2 $V --check
run $N/test_sp_ag_attention_inter_node.py $V --check --seqlens 512 1024
run $A/test_wait_and_notify.py $V
run $A/test_libdevice.py
run $A/test_rocshmem_api.py $V
run $A/test_matmul_amd.py --check
run $A/test_matmul_amd.py -M 4096 -N 8192 -K 8192 --iters 20
run $A/test_extern_functions.py
run $A/test_hip_lds.py
G=triton_dist/mega_triton_kernel/test
run $G/ops/test_add.py
run $G/ops/test_rms_norm.py
run $G/ops/test_mlp_layer.py
run $G/ops/test_page_attn.py
run $G/ops/test_allreduce.py 4
run $G/models/test_qwen3.py 1
run $G/models/test_qwen3.py 2 graph
grep -E "^== |^-- rc|✅" "$out" | paste - - - 2>/dev/null | grep -v "rc=0" | head -40
echo "passed: $(grep -c '^-- rc=0' "$out") of $(grep -c '^== ' "$out")"
