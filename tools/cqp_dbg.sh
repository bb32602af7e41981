#!/bin/bash
# usage: tools/cqp_dbg.sh N : QPBAND through CQP -> SBLS -> SLS('gsls') with the backend's debug prints
cd $(dirname $0)/..
python - "$1" <<'PY'
import sys, os, struct, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import problems as P
n, m, H, A, g, c_l, c_u, x_l, x_u = P.qpband(int(sys.argv[1]))
with open("/tmp/cqp_p.bin", "wb") as f:
    f.write(struct.pack("<2i", 1129336146, 1)); f.write(struct.pack("<6i", n, m, len(H[0]), len(A[0]), 4, 0))
    for (r, c, v) in (H, A):
        f.write(np.ascontiguousarray(r, dtype=np.int32).tobytes()); f.write(np.ascontiguousarray(c, dtype=np.int32).tobytes()); f.write(np.ascontiguousarray(v, dtype=np.float64).tobytes())
    for v in (g, c_l, c_u, x_l, x_u): f.write(np.ascontiguousarray(v, dtype=np.float64).tobytes())
PY
ulimit -s unlimited
GSLS_DEBUG=1 ./oracle/_ref/cqp_gsls_driver /tmp/cqp_p.bin /tmp/cqp_r.bin 2>&1 | grep -v "wave stage" | cut -c1-230 | head -${2:-80}
