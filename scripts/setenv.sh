#!/bin/bash
# source scripts/setenv.sh  - environment for running from a checkout (reference: scripts/setenv.sh)
SCRIPT_DIR=$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)
export TDIST_ROOT=$(realpath "${SCRIPT_DIR}/..")
export PYTHONPATH="${TDIST_ROOT}:${PYTHONPATH:-}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
export PYTORCH_ROCM_ARCH=gfx950
export TDIST_OFFLOAD_ARCH=gfx950
