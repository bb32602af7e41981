"""Randomised soak of the tree-sharded path: both ranks (one GPU, gloo) draw the same seeded systems, factorize three
times and solve; rank 0 compares with numpy.  python -m torch.distributed.run --nproc-per-node 2 tools/soak_shard.py N seed"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, torch.distributed as dist
import soak as SK          # (its main() is guarded below by the import check)
from galahad_amd import SLS, SMT, Control, InformSLS
from galahad_amd.shard import TreeShardedSLS

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    rng = np.random.default_rng(seed)
    bad = 0; ran = 0
    for it in range(N):
        kind = ["spd", "indef", "saddle", "weakdiag"][it % 4]
        n = int(rng.integers(150, 1500))
        A = SK.make(rng, kind, n)
        ev = np.linalg.eigvalsh(A)
        if np.abs(ev).min() < 1e-8 * np.abs(ev).max():
            continue
        r, c = np.nonzero(np.tril(A))
        row, col, val = (r + 1).astype(np.int32), (c + 1).astype(np.int32), A[r, c]
        xs = rng.uniform(-1, 1, n); rhs = A @ xs
        nem = int(rng.choice([4, 8, 16, 24, 32]))
        m = SMT(n, "COORDINATE", row=row, col=col, val=val)
        s, ctl, i = SLS(), Control(), InformSLS(); s.initialize("gsls", ctl, i)
        posdef = kind == "spd" and it % 8 < 4
        ctl.pivot_control = 2 if posdef else 1
        ctl.node_amalgamation = nem
        s.analyse(m, ctl, i)
        d_val = torch.from_numpy(s.scatter_values(m)).cuda()
        try:
            ts = TreeShardedSLS(s, d_val=None if posdef else d_val)
        except Exception as e:        # trees too small to deal
            s.terminate(); continue
        ran += 1
        xd = np.linalg.solve(A, rhs); cond = np.abs(ev).max() / np.abs(ev).min()
        for rep in range(3):
            st = ts.factorize_dev(d_val, posdef)
            ok = st["flag"] >= 0
            err = -1.0
            if ok:
                d_x = torch.from_numpy(rhs.copy()).cuda()
                ts.solve_dev(d_x)
                x = d_x.cpu().numpy()
                err = np.abs(x - xd).max() / max(1.0, np.abs(xd).max())
                ok = (err <= 1e-11 * max(cond, 1e2)) and st["num_neg"] == int((ev < 0).sum()) and st["matrix_rank"] == n
            if not ok:
                bad += 1
                if rank == 0:
                    print("FAIL it %d kind %s n %d nemin %d rep %d flag %d neg %d/%d err %.2e cond %.1e" % (it, kind, n, nem, rep, st["flag"], st.get("num_neg", -1), int((ev < 0).sum()), err, cond), flush=True)
                break
        s.terminate()
    if rank == 0:
        print("soak_shard: %d systems drawn, %d sharded and checked, %d failures" % (N, ran, bad), flush=True)
    dist.barrier(); dist.destroy_process_group()
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
