#!/usr/bin/env bash
# Prepare a machine for the end-to-end tests (role of the reference's scripts/build_e2e_env.sh, which
# installs flash-attn / flashinfer and model dependencies).  Nothing has to be installed here: attention,
# rope, norms and the GEMMs are in-tree HIP kernels.  The script builds the two native libraries,
# checks the Python dependencies that the e2e scripts import, and runs the CPU smoke of the model stack.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$ROOT"
source scripts/setenv.sh
python - <<'PY'
import importlib, sys
missing = [m for m in ("torch", "numpy", "transformers", "safetensors") if importlib.util.find_spec(m) is None]
if missing:
    sys.exit(f"missing Python packages: {missing} (the e2e scripts need them; no network install is attempted)")
import torch
print(f"torch {torch.__version__}, HIP {torch.version.hip}, GPUs visible: {torch.cuda.device_count()}")
PY
python -m triton_distributed_gh_amd.build
RANDOM_PARAMS=1 python triton_dist/test/amd/test_e2e_inference.py --model tdist/tiny-dense --bsz 2 --gen_len 4 --max_length 64 "$@"
echo "e2e environment ready: bash scripts/launch_amd.sh triton_dist/test/amd/test_tp_e2e.py --model Qwen/Qwen3-32B --check"
