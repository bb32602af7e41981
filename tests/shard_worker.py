"""Worker for tests/test_shard_gpu.py: one rank of a tree-sharded factorize+solve.  Launched as
    RANK=r WORLD_SIZE=w MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/shard_worker.py <case> <out.json>
All ranks share cuda:0 (the GPU box has one card); the exchange goes over gloo with host staging,
the device work is the same libgsls.so entry points an RCCL run uses."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402


def problem(case):
    import problems as P
    if case == "grid2d_spd":
        return P.grid2d(60, 50), True
    if case == "banded_spd":
        return P.banded_spd(20000, 31, seed=7), True
    if case == "grid3d_spd":
        return P.grid3d(14, 13, 12), True
    if case == "grid3d_ldlt":              # SPD matrix through the pivoted LDL^T kernels
        return P.grid3d(12, 11, 10), False
    if case == "cfg5_ldlt":                # BASELINE.json configs[4] pattern (27-point + long-range edges), reduced grid
        return P.grid3d_27pt_perturbed(36, 36, 36), False
    if case == "random_indef":             # mixed-sign diagonal, 2x2 pivots and a few delays
        return P.random_sparse(4000, 6, seed=11, spd=False), False
    if case in ("kkt_indef", "kkt_refined"):   # saddle point: delayed pivots -> sharded order repair, unless the
        return P.kkt_qpband(3000, 600, seed=3), False      # order is refined with the values first (kkt_refined)
    if case == "grid2d_indef":
        return P.grid2d(60, 50, shift=1.0), False
    raise SystemExit("unknown case " + case)


def main():
    case, out = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from galahad_amd import SLS, SMT, Control, InformSLS
    from galahad_amd.shard import TreeShardedSLS
    import problems as P

    (n, row, col, val, rhs, xs), posdef = problem(case)
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)

    def analysed():
        s, c, inf = SLS(), Control(), InformSLS()
        s.initialize("gsls", c, inf)
        c.pivot_control = 2 if posdef else 1
        s.analyse(m, c, inf)
        assert inf.status == 0, inf.status
        return s, c, inf

    # single-device answer on this rank (the comparison target)
    s1, c1, i1 = analysed()
    s1.factorize(m, c1, i1)
    assert i1.status == 0, i1.status
    c1.max_iterative_refinements = 0
    x1 = s1.solve(m, rhs.copy(), c1, i1)

    s2, c2, i2 = analysed()
    d_val = torch.from_numpy(s2.scatter_values(m)).cuda()
    ts = TreeShardedSLS(s2, d_val=d_val if case == "kkt_refined" else None)
    owner, cut = ts.partition()
    st = ts.factorize_dev(d_val, posdef)
    d_x = torch.from_numpy(rhs.copy()).cuda()
    ts.solve_dev(d_x)
    x2 = d_x.cpu().numpy()
    res = {"rank": rank, "flag": st["flag"], "num_neg": st["num_neg"], "num_two": st["num_two"],
           "matrix_rank": st["matrix_rank"], "num_delay": st["num_delay"], "ref_num_neg": i1.negative_eigenvalues,
           "ref_two": i1.two_by_two_pivots, "ref_delays": i1.delayed_pivots,
           "max_abs_diff_vs_single": float(np.abs(x2 - x1).max()),
           "bitwise_equal": bool(np.array_equal(x2, x1)),
           "scaled_residual": float(P.scaled_residual(n, row, col, val, x2, rhs)),
           "err_vs_exact": float(np.abs(x2 - xs).max()),
           "owners": sorted(set(int(o) for o in owner)), "ncut": int(len(cut)),
           "xsum": float(x2.sum())}
    # a second and a third factorize+solve on the same handles: the refactorization path (when the first pass needed no
    # pivoting both the single-device handle and the sharded one switch the tiny fronts to the wave-per-front kernels:
    # the sharded result must be bitwise the single-device refactorization's), and its repeat
    s1.factorize(m, c1, i1)
    x1b = s1.solve(m, rhs.copy(), c1, i1)
    st = ts.factorize_dev(d_val, posdef)
    d_x = torch.from_numpy(rhs.copy()).cuda()
    ts.solve_dev(d_x)
    x2b = d_x.cpu().numpy()
    st = ts.factorize_dev(d_val, posdef)
    d_x = torch.from_numpy(rhs.copy()).cuda()
    ts.solve_dev(d_x)
    x2c = d_x.cpu().numpy()
    res["refactor_bitwise_vs_single"] = bool(np.array_equal(x2b, x1b))
    res["refactor_max_abs_diff_vs_single"] = float(np.abs(x2b - x1b).max())
    res["refactor_residual"] = float(P.scaled_residual(n, row, col, val, x2b, rhs))
    res["repeat_bitwise"] = bool(np.array_equal(x2c, x2b))
    with open(out + ".%d" % rank, "w") as f:
        json.dump(res, f)
    s1.terminate()
    s2.terminate()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
