#!/bin/bash
mkdir -p gpurun_out/r3
for v in "0 0" "2 0" "0 2" "2 2"; do
  set -- $v
  GSLS_EXTRA="-DGSLS_WS_NT_F=$1 -DGSLS_WS_NT_B=$2" bash galahad_amd/csrc/build.sh > /dev/null 2>&1
  echo "nt fwd $1 bwd $2"
  timeout -k 10 150 bash tools/prof_bench.sh r3nt > gpurun_out/r3/prof_nt.txt 2>&1
  python tools/last_step.py gpurun_out/prof_r3nt x 2>&1 | grep -E "^void gsls::k_wsolve|last step"
  tail -2 gpurun_out/r3/prof_nt.txt
done
bash galahad_amd/csrc/build.sh > /dev/null 2>&1
