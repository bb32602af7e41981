#!/bin/bash
GSLS_EXTRA=-DGSLS_STAMPS bash galahad_amd/csrc/build.sh > /dev/null 2>&1
python tools/stamp_tail.py
bash galahad_amd/csrc/build.sh > /dev/null 2>&1
