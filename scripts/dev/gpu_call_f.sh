#!/bin/bash
# kernel-trace and two counter passes over the solo-rank probe at w = 1 (fused kernels vs plain GEMM)
set -o pipefail
O=gpurun_out/r2/solo1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python3 $R/scripts/dev/probe_solo_rank.py 1 > $R/$O/kt.log 2>&1
rc=$?; tail -n 3 $R/$O/kt.log; [ $rc -ne 0 ] && exit $rc
python3 $R/scripts/rocpd_kernel_stats.py $(find $R/$O/kt -name '*results.db' | head -1) --skip-first 8 > $R/$O/kernel_stats.txt 2>&1
head -n 12 $R/$O/kernel_stats.txt
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -d $R/$O/pmc1 -o p -- python3 $R/scripts/dev/probe_solo_rank.py 1 > $R/$O/pmc1.log 2>&1
rc=$?; tail -n 3 $R/$O/pmc1.log; [ $rc -ne 0 ] && exit $rc
python3 $R/scripts/rocpd_pmc_summary.py $(find $R/$O/pmc1 -name '*results.db' | head -1) --skip-first 8 > $R/$O/pmc1.txt 2>&1
cat $R/$O/pmc1.txt | cut -c1-400
find $R/$O -name '*.db' -delete
