#!/bin/bash
# diagnostic: calibrate the in-kernel stamp unit against rocprofv3 kernel durations
set -e
GSLS_EXTRA=-DGSLS_STAMPS bash galahad_amd/csrc/build.sh >/dev/null
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_stamp -- python3 $R/tools/stamp_run.py 256 2>&1 | grep "ns:" | tail -1
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_stamp/*/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_diag_chol' in r['Kernel_Name']]
for r in rows[-4:]:
    print('k_diag_chol duration ns', int(r['End_Timestamp'])-int(r['Start_Timestamp']))
PY
