#!/bin/bash
# flash decode: throughput lines and two counter passes
set -o pipefail
R=$GRAFT_REPO_ROOT
O=gpurun_out/fdec
mkdir -p $R/$O
T=$R/triton_dist/test/nvidia/test_decode_attn.py
timeout -k 10 200 python -u $T --batch 64 --kv-len 4096 --page-size 64 > $R/$O/b64.log 2>&1 || exit 1
tail -n 3 $R/$O/b64.log
timeout -k 10 200 python -u $T --batch 1 --kv-len 32768 --page-size 64 > $R/$O/b1.log 2>&1 || exit 1
tail -n 3 $R/$O/b1.log
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace -d $R/$O/p1 -o p -- python3 $T --batch 64 --kv-len 4096 --page-size 64 > $R/$O/p1.log 2>&1 || exit 1
python3 $R/scripts/rocpd_pmc_summary.py $(find $R/$O/p1 -name '*results.db' | head -1) --match decode_split > $R/$O/pmc1.txt 2>&1
cat $R/$O/pmc1.txt | cut -c1-500
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/p2 -o p -- python3 $T --batch 64 --kv-len 4096 --page-size 64 > $R/$O/p2.log 2>&1 || exit 1
python3 $R/scripts/rocpd_pmc_summary.py $(find $R/$O/p2 -name '*results.db' | head -1) --match decode_split > $R/$O/pmc2.txt 2>&1
cat $R/$O/pmc2.txt | cut -c1-300
find $R/$O -name '*.db' -delete
