#!/bin/bash
# rehearse the default N>1 mode of bench.py (one independent system per rank) with W ranks sharing the one
# GPU of a gpurun box (gloo); timing is NOT meaningful -- this checks the code path end to end
W=${1:-2}
PORT=${2:-29543}
pids=()
for r in $(seq 0 $((W-1))); do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=$W MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT \
    python bench.py --gpus $W --steps 5 --warmup 2 --backend gloo --no-cpu-baseline &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
exit $rc
