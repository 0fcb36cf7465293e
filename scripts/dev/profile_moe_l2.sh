#!/bin/bash
# L2 / HBM traffic of the grouped GEMMs (solo rank, Mixtral shapes) against dense GEMMs of the same size.
# Two counter passes per program (TCC hit / miss / requests; FETCH_SIZE).  Writes gpurun_out/moe_l2/*.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/moe_l2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for prog in probe_moe_solo probe_moe_dense_equiv; do
  NCK=1 ITERS=4 timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $O/${prog}_a -o p -- python3 $R/scripts/dev/$prog.py > $O/${prog}_a.log 2>&1 || { tail -5 $O/${prog}_a.log; exit 1; }
  python3 $R/scripts/rocpd_pmc_summary.py $(find $O/${prog}_a -name '*results.db' | head -1) > $O/${prog}_tcc.txt 2>&1
  NCK=1 ITERS=4 timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${prog}_b -o p -- python3 $R/scripts/dev/$prog.py > $O/${prog}_b.log 2>&1 || { tail -5 $O/${prog}_b.log; exit 1; }
  python3 $R/scripts/rocpd_pmc_summary.py $(find $O/${prog}_b -name '*results.db' | head -1) > $O/${prog}_fetch.txt 2>&1
  grep -A1 -i "moe_group_gemm\|gemm_nt_256_persistent\|Cijk\|gemm_nt_tile" $O/${prog}_tcc.txt $O/${prog}_fetch.txt | cut -c1-330
done
find $O -name '*.db' -delete
