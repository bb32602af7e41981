#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args]: rocprofv3 kernel stats of a short bench run -> gpurun_out/prof_<tag>
TAG=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-facade "$@" > $R/gpurun_out/bench_$TAG.log 2>&1
cd $R
python tools/kstats.py gpurun_out/prof_$TAG
grep -o "ms_per_step\": [0-9.]*\|seconds_per_launch\": [0-9.e-]*" gpurun_out/bench_$TAG.log
