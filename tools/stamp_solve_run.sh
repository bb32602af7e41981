#!/bin/bash
set -e
GSLS_EXTRA=-DGSLS_STAMPS bash galahad_amd/csrc/build.sh >/dev/null
python tools/stamp_solve.py "$@"
