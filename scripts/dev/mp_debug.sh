#!/bin/bash
# Two (or N) ranks of a script on ONE GPU, started by hand so that a hang ends in Python tracebacks:
# each rank runs under `timeout -s ABRT` with faulthandler enabled.
N=${N:-2}; LIMIT=${LIMIT:-90}; PORT=${PORT:-29533}
mkdir -p gpurun_out
pids=()
for r in $(seq 0 $((N-1))); do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=$N LOCAL_WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT PYTHONFAULTHANDLER=1 \
    timeout -s ABRT -k 10 $LIMIT python -u -X faulthandler "$@" > gpurun_out/mp_rank$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
for r in $(seq 0 $((N-1))); do echo "---- rank $r"; tail -n 40 gpurun_out/mp_rank$r.log; done
exit $rc
