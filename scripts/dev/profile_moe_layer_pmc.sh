#!/bin/bash
# HBM fetch size and L2 hit / miss of the grouped MoE kernels at a token count (one Qwen3-30B-A3B layer, one rank).
# Two counter passes (TCC_HIT / MISS / REQ / EA0_RDREQ; FETCH_SIZE).  usage: profile_moe_layer_pmc.sh <tokens>
set -o pipefail
tok=${1:-64}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3/moe_layer_pmc_$tok
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export NO_TORCH=1 CTX_M=8192 TOKENS=$tok
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $O/a -o p -- python3 $R/scripts/dev/probe_moe_layer_tokens.py > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
python3 $R/scripts/rocpd_pmc_summary.py $(find $O/a -name '*results.db' | head -1) > $O/tcc.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/b -o p -- python3 $R/scripts/dev/probe_moe_layer_tokens.py > $O/b.log 2>&1 || { tail -5 $O/b.log; exit 1; }
python3 $R/scripts/rocpd_pmc_summary.py $(find $O/b -name '*results.db' | head -1) > $O/fetch.txt 2>&1
grep -A1 -i "moe_group16\|moe_group_mid\|moe_group_gemm" $O/tcc.txt $O/fetch.txt | cut -c1-300
find $O -name '*.db' -delete
