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


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus N` launched WITHOUT torchrun (the way the driver launches --gpus 1) must run N ranks:
    the parent spawns N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and never initialises torch or
    HIP itself (VERDICT r2 weak #5a).  GSLS_BENCH_SPAWN_ONLY makes every rank report its environment and stop."""
    import json
    import subprocess
    env = dict(os.environ, GSLS_BENCH_SPAWN_ONLY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")), key=lambda d: d["rank"])
    assert [d["rank"] for d in got] == [0, 1, 2] and [d["local_rank"] for d in got] == [0, 1, 2]
    assert all(d["world"] == 3 for d in got)
    assert len({d["master"] for d in got}) == 1 and got[0]["master"].startswith("127.0.0.1:")
    # under a launcher (WORLD_SIZE set) nothing is spawned: this process IS a rank
    env2 = dict(env, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True,
                       text=True, timeout=120)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1 and lines[0]["rank"] == 1 and lines[0]["world"] == 2


def test_in_library_communicator_argument_checking(monkeypatch, tmp_path):
    """The RCCL transport inside libgsls.so (gsls_comm_*, what a Fortran host binds: galahad_amd/fortran/gsls_iface.f90)
    cannot run on this box -- but every way of calling it wrongly must come back as a flag, before any device or
    communicator call (VERDICT r2 #2e).  Also the environment-driven form that SLS callers use (gsls_comm_init_env)."""
    import ctypes as C
    import problems as P
    from galahad_amd import SLS, SMT, Control, InformSLS
    from galahad_amd._lib import Inform, lib
    ident = b"\\0" * 128
    inf = Inform()
    assert lib.gsls_comm_init(None, 2, 0, ident, None) == -1
    n, row, col, val, rhs, xs = P.grid2d(12, 12)
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    h = s.handle
    assert lib.gsls_comm_init(h, 2, 0, ident, C.byref(s.opts)) == -1                 # not analysed yet
    s.analyse(SMT(n, "COORDINATE", row=row, col=col, val=val), c, i)
    assert i.status == 0
    for nranks, rank, idp in ((1, 0, ident), (2, 2, ident), (2, -1, ident), (2, 0, None)):
        assert lib.gsls_comm_init(h, nranks, rank, idp, C.byref(s.opts)) == -1, (nranks, rank)
    # without a communicator every collective entry point refuses, and leaves a zeroed inform with the flag
    x = np.zeros(n)
    for f in (lib.gsls_comm_factor(h, 1, x.ctypes.data_as(C.c_void_p), C.byref(s.opts), C.byref(inf)),
              lib.gsls_comm_factor_dev(h, 1, None, C.byref(s.opts), C.byref(inf)),
              lib.gsls_comm_solve(h, x.ctypes.data_as(C.c_void_p), C.byref(inf)),
              lib.gsls_comm_solve_dev(h, None, C.byref(inf)),
              lib.gsls_comm_collect_dev(h, None, C.byref(inf))):
        assert f == -1 and inf.flag == -1
    assert lib.gsls_comm_destroy(h) == 0 and lib.gsls_comm_destroy(None) == 0
    # the environment-driven form: nothing set -> nothing happens; one rank -> nothing; a rank out of range -> flag
    for k in ("GSLS_COMM_RANKS", "GSLS_COMM_RANK", "GSLS_COMM_ID_FILE"):
        monkeypatch.delenv(k, raising=False)
    assert lib.gsls_comm_init_env(h, C.byref(s.opts)) == 0
    monkeypatch.setenv("GSLS_COMM_RANKS", "1")
    monkeypatch.setenv("GSLS_COMM_RANK", "0")
    monkeypatch.setenv("GSLS_COMM_ID_FILE", str(tmp_path / "id"))
    assert lib.gsls_comm_init_env(h, C.byref(s.opts)) == 0
    monkeypatch.setenv("GSLS_COMM_RANKS", "4")
    monkeypatch.setenv("GSLS_COMM_RANK", "4")
    assert lib.gsls_comm_init_env(h, C.byref(s.opts)) == -1
    s.terminate()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_per_rank_memory_of_a_sharded_run(world):
    """VERDICT r2 weak #5d: a rank of a tree-sharded run allocates the factors of the fronts it OWNS (rank 0: and the top
    part) and a lifetime-reused arena for their contribution blocks -- not the whole matrix.  Host logic, no device:
    gsls_get_layout_sizes after gsls_shard, on the BASELINE configs[4] pattern (27-point 3-D grid) at 20^3."""
    import ctypes as C
    import problems as P
    from galahad_amd import SLS, SMT, Control, InformSLS
    from galahad_amd._lib import lib
    n, row, col, val = P.grid3d(20, 20, 20)[:4]
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)

    def handle():
        s, c, inf = SLS(), Control(), InformSLS()
        s.initialize("gsls", c, inf)
        s.analyse(m, c, inf)
        assert inf.status == 0
        return s

    s0 = handle()
    fe, ae = C.c_int64(), C.c_int64()
    assert lib.gsls_get_layout_sizes(s0.handle, C.byref(fe), C.byref(ae)) == 0
    whole_factor, whole_arena = fe.value, ae.value
    sym = s0.symbolic()
    nn = sym["nnodes"]
    ncol, nrow = np.diff(sym["sptr"]).astype(np.int64), np.diff(sym["rptr"]).astype(np.int64)
    size = ((nrow + 1) // 2 * 2) * ncol                       # ld (rows rounded to even) x columns, per front
    assert whole_factor == int(size.sum())
    side_by_side = int(((nrow - ncol) ** 2).sum())             # what every rank allocated before: all blocks, no reuse
    per_rank = []
    for r in range(world):
        s = handle()
        assert lib.gsls_shard(s.handle, world, r, None, None) == 0
        owner = np.zeros(nn, dtype=np.int32)
        lib.gsls_shard_get(s.handle, owner.ctypes.data_as(C.POINTER(C.c_int32)), None, None)
        assert lib.gsls_get_layout_sizes(s.handle, C.byref(fe), C.byref(ae)) == 0
        mine = (owner == r) | ((owner < 0) & (r == 0))
        assert fe.value == int(size[mine].sum())                # exactly the owned fronts
        per_rank.append((fe.value, ae.value))
        s.terminate()
    assert sum(f for f, _ in per_rank) == whole_factor          # every front on exactly one rank
    for r, (f, a) in enumerate(per_rank):
        assert a < side_by_side                                 # lifetime reuse, this rank's blocks only
        if r > 0:
            assert f < whole_factor // 2                        # a subtree owner holds a fraction of L
    # going back to one device restores the single-device layout
    assert lib.gsls_shard(s0.handle, 1, 0, None, None) == 0
    assert lib.gsls_get_layout_sizes(s0.handle, C.byref(fe), C.byref(ae)) == 0
    assert fe.value == whole_factor and ae.value == whole_arena
    s0.terminate()
