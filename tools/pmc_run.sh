#!/bin/bash
# usage: tools/pmc_run.sh <tag> [bench args]
# two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), each with --kernel-trace only, as
# MI355X_MICROARCH.md prescribes; results -> gpurun_out/pmc_<tag>_{fetch,write}/ and a JSON summary
TAG=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-facade "$@" > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1 || exit 1
done
cd $R
python3 tools/pmc_traffic.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE "$TAG $*" > gpurun_out/pmc_traffic_$TAG.json
cat gpurun_out/pmc_traffic_$TAG.json
