import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as P
from oracle import refio
n, row, col, val, rhs, xs = P.kkt_qpband(1000000, 200000)
os.environ["GSLS_DEBUG"] = "1"
import subprocess
# run the driver directly to see stderr
import tempfile
d = tempfile.mkdtemp()
pin, pout = os.path.join(d, "p.bin"), os.path.join(d, "r.bin")
refio.write_problem(pin, n, row, col, val, rhs, solver="gsls", pivot_control=1, nemin=24, repeat=3, max_refine=1)
p = subprocess.run(["bash", "-c", "ulimit -s unlimited; exec '%s' '%s' '%s'" % (refio.DROPIN, pin, pout)], capture_output=True, text=True, env=dict(os.environ, OMP_CANCELLATION="true"))
print(p.stderr[-3000:])
r = refio.read_result(pout, n, 1, False)
print("factorize median %.2f solve median %.2f ms" % (r["t_factorize_median"] * 1e3, r["t_solve_median"] * 1e3))
