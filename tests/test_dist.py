"""CPU, world_size 2, gloo: the rank bookkeeping bench.py uses for N>1 (unit partition, barrier,
max-over-ranks time, whole-job aggregate)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from galahad_amd import dist as gdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    units = list(gdist.shard_units(7, world, rank))
    gdist.barrier(world)
    tmax = gdist.max_over_ranks(1.0 + rank, world)
    total = gdist.sum_over_ranks(len(units), world)
    q.put((rank, units, tmax, total))
    dist.destroy_process_group()


def test_two_rank_partition_and_reductions():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2, 3] and res[1][1] == [4, 5, 6]      # disjoint, covering, balanced
    assert res[0][2] == res[1][2] == 2.0                             # max over ranks
    assert res[0][3] == res[1][3] == 7.0                             # whole-job units


def test_shard_units_edge_cases():
    sys.path.insert(0, ROOT)
    from galahad_amd.dist import shard_units
    assert [list(shard_units(2, 4, r)) for r in range(4)] == [[0], [1], [], []]
    assert list(shard_units(0, 3, 1)) == []
    assert sum(len(shard_units(1001, 8, r)) for r in range(8)) == 1001


# ---- elimination-tree sharding: the partition (host integer work, no GPU) -----------------------------
@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("case", ["grid2d", "band", "grid3d"])
def test_tree_partition_invariants(case, world):
    """gsls_shard (anal.f90:284-459 recipe): owned parts are whole subtrees, the top part is closed
    under 'parent', cut roots are exactly the owned nodes whose parent is in the top part, and the
    subtree work is spread over all ranks."""
    import ctypes as C
    import problems as P
    from galahad_amd import SLS, SMT, Control, InformSLS
    from galahad_amd._lib import lib
    prob = {"grid2d": lambda: P.grid2d(60, 50), "band": lambda: P.banded_spd(20000, 31, seed=7),
            "grid3d": lambda: P.grid3d(14, 13, 12)}[case]()
    n, row, col, val = prob[:4]
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, c, inf = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, inf)
    s.analyse(m, c, inf)
    assert inf.status == 0
    ce, ve = C.c_int64(), C.c_int64()
    assert lib.gsls_shard(s.handle, world, 0, C.byref(ce), C.byref(ve)) == 0
    sym = s.symbolic()
    nn = sym["nnodes"]
    owner = np.zeros(nn, dtype=np.int32)
    ncut = C.c_int32()
    lib.gsls_shard_get(s.handle, owner.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ncut), None)
    cut = np.zeros(max(ncut.value, 1), dtype=np.int32)
    lib.gsls_shard_get(s.handle, None, None, cut.ctypes.data_as(C.POINTER(C.c_int32)))
    cut = cut[: ncut.value] - 1
    parent = sym["sparent"] - 1          # nn = virtual root
    assert owner.min() >= -1 and owner.max() < world
    expect_cut = []
    for k in range(nn):
        p = parent[k]
        if owner[k] < 0:
            assert p >= nn or owner[p] < 0              # top part closed upwards
        elif p < nn:
            if owner[p] < 0:
                expect_cut.append(k)
            else:
                assert owner[p] == owner[k]             # owned parts are whole subtrees
    assert sorted(expect_cut) == sorted(cut.tolist())
    ncol = np.diff(sym["sptr"])
    nrow = np.diff(sym["rptr"])
    cm = (nrow - ncol)[cut].astype(np.int64)
    assert ce.value == int((cm * cm).sum()) + 24          # the cut roots' blocks + three status blocks of 8
    assert ve.value == max(int(cm.sum()), n)
    assert set(range(world)) <= set(owner.tolist())     # every rank got a subtree
    # same call on another "rank" gives the same partition (every process computes it independently)
    s2, c2, i2 = SLS(), Control(), InformSLS()
    s2.initialize("gsls", c2, i2)
    s2.analyse(m, c2, i2)
    assert lib.gsls_shard(s2.handle, world, world - 1, None, None) == 0
    owner2 = np.zeros(nn, dtype=np.int32)
    lib.gsls_shard_get(s2.handle, owner2.ctypes.data_as(C.POINTER(C.c_int32)), None, None)
    assert np.array_equal(owner, owner2)
    s.terminate()
    s2.terminate()
