import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, problems as P
from oracle import refio
for N in [int(a) for a in sys.argv[2:]] or [5]:
    r = refio.run_cqp(*P.qpband(N), solver=sys.argv[1], print_level=int(os.environ.get("PL", "0")))
    print(N, sys.argv[1], "status", r["status"], "iter", r["iter"], "nfacts", r["nfacts"], "obj %.10e" % r["obj"], "pf %.1e df %.1e cs %.1e" % (r["primal_infeasibility"], r["dual_infeasibility"], r["complementary_slackness"]), "time %.3f (factorize %.3f solve %.3f)" % (r["time_total"], r["time_factorize"], r["time_solve"]), flush=True)
