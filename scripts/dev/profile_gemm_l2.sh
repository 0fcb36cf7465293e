#!/bin/bash
# L2 traffic of the native GEMM against the library kernel at the headline shape
set -o pipefail
R=$GRAFT_REPO_ROOT
O=gpurun_out/gemm_l2
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 100 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $R/$O/p1 -o p -- python3 $R/scripts/dev/run_gemm_pair_once.py > $R/$O/p1.log 2>&1 || { tail -5 $R/$O/p1.log; exit 1; }
python3 $R/scripts/rocpd_pmc_summary.py $(find $R/$O/p1 -name '*results.db' | head -1) > $R/$O/pmc1.txt 2>&1
grep -i "gemm_nt_256\|Cijk\|kernel" $R/$O/pmc1.txt | cut -c1-400
timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/p2 -o p -- python3 $R/scripts/dev/run_gemm_pair_once.py > $R/$O/p2.log 2>&1 || { tail -5 $R/$O/p2.log; exit 1; }
python3 $R/scripts/rocpd_pmc_summary.py $(find $R/$O/p2 -name '*results.db' | head -1) > $R/$O/pmc2.txt 2>&1
grep -i "gemm_nt_256\|Cijk\|kernel" $R/$O/pmc2.txt | cut -c1-400
find $R/$O -name '*.db' -delete
