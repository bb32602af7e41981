#!/bin/bash
# rehearse `bench.py --shard tree` with W ranks sharing the one GPU of a gpurun box (exchange over gloo);
# timing is NOT meaningful (the ranks share the card) -- this checks the code path end to end
W=${1:-2}
PORT=${2:-29541}
pids=()
for r in $(seq 0 $((W-1))); do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=$W MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT \
    python bench.py --gpus $W --steps 5 --warmup 2 --shard tree --backend gloo --no-cpu-baseline &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
exit $rc
