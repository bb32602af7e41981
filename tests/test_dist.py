"""CPU, world_size 2, gloo: the rank bookkeeping bench.py uses for N>1 (unit partition, barrier,
max-over-ranks time, whole-job aggregate)."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
