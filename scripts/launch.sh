#!/bin/bash
# Launch a script on every GPU of this node, one process per GPU, through torchrun.
#   bash scripts/launch.sh python/triton_dist/test/amd/test_ag_gemm_intra_node.py 8192 11008 4096
# Role in the reference: scripts/launch.sh:128-175 (the NVIDIA launcher; same launcher here, there is one platform) (PYTHONPATH set-up, HIP env quirks, torchrun).
SCRIPT_DIR=$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)
REPO_ROOT=$(realpath "${SCRIPT_DIR}/..")
export PYTHONPATH="${REPO_ROOT}:${PYTHONPATH}"

# dmabuf IPC is the only IPC mode the MI355X host driver stack supports
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
# a rank's compute stream and its side (push / copy) streams must not share a hardware queue
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
export TORCH_NCCL_ASYNC_ERROR_HANDLING=${TORCH_NCCL_ASYNC_ERROR_HANDLING:-1}
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-8}

nproc_per_node=${ARNOLD_WORKER_GPU:-${NPROC_PER_NODE:-$(python3 -c "import torch; print(max(torch.cuda.device_count(), 1))")}}
nnodes=${ARNOLD_WORKER_NUM:-1}
node_rank=${ARNOLD_ID:-0}
master_addr=${ARNOLD_WORKER_0_HOST:-127.0.0.1}
master_port=${ARNOLD_WORKER_0_PORT:-${MASTER_PORT:-23456}}
if [ "${nnodes}" != "1" ]; then
  echo "this framework targets one xGMI node (nnodes=${nnodes} requested)" >&2
  exit 2
fi

CMD="python3 -m torch.distributed.run --nnodes=1 --node-rank=${node_rank} --nproc-per-node=${nproc_per_node} \
  --master-addr=${master_addr} --master-port=${master_port} ${DIST_TRITON_EXTRA_TORCHRUN_ARGS} $*"
echo "${CMD}"
exec ${CMD}
