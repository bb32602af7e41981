#!/bin/bash
# usage (on the GPU box, via gpurun): tools/measure_round.sh r03
# everything the round's measurement section cites, into gpurun_out/<round>/ (copied to profiles/<round>/ afterwards):
#   kernel-trace stats + last-step breakdown (kkt, band), FETCH_SIZE / WRITE_SIZE passes (kkt, band), the
#   MFMA / SQ counter passes, the FETCH_SIZE calibration, the bench lines themselves
RND=${1:-r03}
R=$PWD
O=$R/gpurun_out/$RND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for WL in kkt band; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$WL -- python3 $R/bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-facade > $O/trace_bench_$WL.log 2>&1
  (cd $R && python3 tools/kstats.py $O/trace_$WL > $O/${WL}_summary.txt && python3 tools/last_step.py $O/trace_$WL x > $O/${WL}_last_step.txt)
  cp $(ls -t $O/trace_$WL/*/*_kernel_stats.csv | head -1) $O/${WL}_kernel_stats.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${WL}_$C -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-facade > $O/pmc_${WL}_$C.log 2>&1
  done
  (cd $R && python3 tools/pmc_traffic.py $O/pmc_${WL}_FETCH_SIZE $O/pmc_${WL}_WRITE_SIZE "$WL" > $O/pmc_traffic_$WL.json)
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_${WL}_SQ -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-facade > $O/pmc_${WL}_SQ.log 2>&1
  (cd $R && python3 tools/pmc_sum.py $O/pmc_${WL}_SQ > $O/pmc_sq_$WL.txt && python3 tools/mfma_util.py $O/pmc_sq_$WL.txt > $O/mfma_util_$WL.txt)
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${WL}_GRBM -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-facade > $O/pmc_${WL}_GRBM.log 2>&1
  (cd $R && python3 tools/pmc_sum.py $O/pmc_${WL}_GRBM > $O/pmc_grbm_$WL.txt)
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -- $R/tools/ubench_calib > $O/calib.log 2>&1
(cd $R && python3 tools/pmc_sum.py $O/calib > $O/calib_fetch.txt)
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench_kkt.json 2> $O/bench_kkt.err
python3 bench.py --workload band --steps 20 --warmup 5 > $O/bench_band.json 2> $O/bench_band.err
ls $O
