#!/usr/bin/env bash
# register / LDS / occupancy report of the kernels whose name matches $1 (default: all)
cd "$(dirname "$0")/../galahad_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c gsls_device.hip -o /tmp/dev_kres.o -Rpass-analysis=kernel-resource-usage 2>/tmp/kres.txt
python3 - "$1" <<'PY'
import re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
txt = open('/tmp/kres.txt').read()
if ' error' in txt: print(txt[:3000])
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split(' ')[0]
    if pat and not re.search(pat, name): continue
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    print("%-70s VGPR %3s SGPR %3s spill s%s v%s LDS %6s occ %s" % (name[:70], g('VGPRs'), g('TotalSGPRs'), g('SGPRs Spill'), g('VGPRs Spill'), g(r'LDS Size \[bytes/block\]'), g(r'Occupancy \[waves/SIMD\]')))
PY
