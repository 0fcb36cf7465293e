#!/bin/bash
# Build every native library in-tree for gfx950 (works without a GPU).
SCRIPT_DIR=$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)
cd "${SCRIPT_DIR}/.." && python3 -m triton_distributed_gh_amd.build
