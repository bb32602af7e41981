#!/bin/bash
# diagnostic: build libgsls.so with in-kernel stamps into a scratch copy, run tools/stamp_run.py
set -e
GSLS_EXTRA=-DGSLS_STAMPS bash galahad_amd/csrc/build.sh >/dev/null
python tools/stamp_run.py "$@"
