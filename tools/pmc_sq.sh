#!/bin/bash
# usage: tools/pmc_sq.sh <tag> "<counters>" [bench args]: one rocprofv3 --pmc pass of a short bench run;
# prints per-kernel sums of the counters over the LAST step's dispatches of every kernel -> gpurun_out/pmc_<tag>.txt
TAG=$1; CNT=$2; shift; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-facade "$@" > $R/gpurun_out/pmcbench_$TAG.log 2>&1
cd $R
python3 tools/pmc_sum.py gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG.txt
cat gpurun_out/pmc_$TAG.txt
