#!/bin/bash
# Regenerate the AOT manifest + registration library from scripts/aot_kernels.txt
# (reference: scripts/gen_aot_code.sh).
SCRIPT_DIR=$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)
export PYTHONPATH="${SCRIPT_DIR}/..:${PYTHONPATH}"
KERNELS=$(grep -v '^#' "${SCRIPT_DIR}/aot_kernels.txt" | grep -v '^$' | SCRIPT_DIR=${SCRIPT_DIR} envsubst 2>/dev/null \
          || grep -v '^#' "${SCRIPT_DIR}/aot_kernels.txt" | grep -v '^$' | sed "s#\${SCRIPT_DIR}#${SCRIPT_DIR}#g")
python3 -m triton_distributed_gh_amd.tools.compile_aot --workspace "${SCRIPT_DIR}/../build/aot" --from-scratch \
  --kernels ${KERNELS} --library triton_distributed_kernel --build
