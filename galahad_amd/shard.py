"""One sparse system factorized across several GPUs: elimination-tree sharding.

The reference deals independent subtrees of the assembly tree to NUMA regions / GPUs
(find_subtree_partition, src/ssids/anal.f90:284-459; assignment :569-590) and keeps the top of the
tree on one owner.  Here that is one process per GPU (torch.distributed, RCCL over xGMI when the
backend is "nccl"): every rank analyses the same matrix (integer work, deterministic, identical on all
ranks), `gsls_shard` deals the subtrees, and the only data exchanged are

  * factorize: the contribution blocks of the subtree roots, summed onto rank 0 (one all-reduce);
  * solve:     their contribution vectors up (one all-reduce), the top part's solution down (one
               broadcast), the assembled solution (one all-reduce).

Every summed element is non-zero on exactly one rank, so results do not depend on reduction order and
are bitwise those of the single-GPU path.  The device work is entirely inside libgsls.so
(gsls_shard_factor_dev / gsls_shard_solve_dev, include/gsls.h); this module only sequences the phases
and the collectives.  With a non-RCCL backend (gloo, used by the tests) the exchange buffers are
staged through host memory.
"""
import ctypes as C

from ._lib import Inform, lib


class TreeShardedSLS:
    """Wraps an analysed galahad_amd.SLS object; all ranks must call every method collectively."""

    def __init__(self, sls, group=None, d_val=None):
        """d_val (device tensor with the matrix values, optional): lets every rank refine its own ordering with
        the values (gsls_refine_order_dev) before the tree is dealt -- same result on all ranks."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.sls = sls
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world < 2:
            raise ValueError("tree sharding needs at least two ranks; use SLS.factorize on one GPU")
        if d_val is not None:
            flag = lib.gsls_refine_order_dev(sls.handle, C.c_void_p(d_val.data_ptr()), None)
            if flag < 0:
                raise RuntimeError("gsls_refine_order_dev failed with flag %d" % flag)
        ce, ve = C.c_int64(), C.c_int64()
        flag = lib.gsls_shard(sls.handle, self.world, self.rank, C.byref(ce), C.byref(ve))
        if flag != 0:
            raise RuntimeError("gsls_shard failed with flag %d" % flag)
        self.n = sls.n
        self.direct = dist.get_backend(group) == "nccl"
        self._alloc(ce.value, ve.value)

    def _cut_vector_elems(self):
        import numpy as np
        sym = self.sls.symbolic()
        _, cut = self.partition()
        return int((np.diff(sym["rptr"])[cut] - np.diff(sym["sptr"])[cut]).sum())

    # ---- collectives on a prefix of a device buffer --------------------------------------------------
    def _sum(self, buf, count):
        if count <= 0:
            return
        t = buf[:count]
        self.torch.cuda.current_stream().synchronize()
        if self.direct:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            self.torch.cuda.current_stream().synchronize()
        else:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            self.torch.cuda.current_stream().synchronize()

    def _bcast0(self, buf, count):
        t = buf[:count]
        src = self.dist.get_global_rank(self.group, 0) if self.group is not None else 0
        if self.direct:
            self.dist.broadcast(t, src=src, group=self.group)
            self.torch.cuda.current_stream().synchronize()
        else:
            h = t.cpu()
            self.dist.broadcast(h, src=src, group=self.group)
            t.copy_(h)
            self.torch.cuda.current_stream().synchronize()

    # ---- SLS_factorize ---------------------------------------------------------------------------------
    def _alloc(self, ce, ve):
        torch = self.torch
        self.factor_elems, self.solve_elems = ce, ve
        dev = torch.device("cuda", torch.cuda.current_device())
        self.xchg_factor = torch.empty(ce, dtype=torch.float64, device=dev)
        self.xchg_solve = torch.empty(ve, dtype=torch.float64, device=dev)
        self.cvec_elems = self._cut_vector_elems()

    def _total(self, value, op=None):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64)
        self._host_reduce(t, op or self.dist.ReduceOp.SUM)
        return t[0].item()

    def _repair(self):
        """all ranks: gather the failed pivots, repair the elimination order identically everywhere"""
        import numpy as np
        nf = C.c_int32()
        mine = np.zeros(16384, dtype=np.int32)
        flag = lib.gsls_shard_failed(self.sls.handle, C.byref(nf), mine.ctypes.data_as(C.POINTER(C.c_int32)))
        if flag != 0:
            raise RuntimeError("gsls_shard_failed: flag %d" % flag)
        lists = [None] * self.world
        self.dist.all_gather_object(lists, mine[: nf.value].tolist(), group=self.group)
        union = np.array(sorted(set(p for part in lists for p in part)), dtype=np.int32)
        ce, ve = C.c_int64(), C.c_int64()
        flag = lib.gsls_shard_repair(self.sls.handle, len(union), union.ctypes.data_as(C.POINTER(C.c_int32)),
                                     C.byref(ce), C.byref(ve))
        if flag != 0:
            return flag, 0
        self._alloc(ce.value, ve.value)
        return 0, len(union)

    def factorize_dev(self, d_val, posdef, max_pass=200):
        """d_val: device tensor holding VAL (SLS.scatter_values) on every rank.  Returns a dict with the
        flag and the statistics summed over ranks."""
        s = self.sls
        moved = 0
        for _ in range(max_pass + 1):
            restart = False
            flags = []
            for phase in (1, 2):
                inf = Inform()
                self.torch.cuda.current_stream().synchronize()
                flag = lib.gsls_shard_factor_dev(s.handle, phase, 1 if posdef else 0,
                                                 C.c_void_p(d_val.data_ptr()),
                                                 C.c_void_p(self.xchg_factor.data_ptr()), C.byref(s.opts),
                                                 C.byref(inf))
                flags.append(flag)
                worst = self._total(min(flags), self.dist.ReduceOp.MIN)
                if worst < 0:
                    return {"flag": int(worst), "num_neg": 0, "num_two": 0, "matrix_rank": 0,
                            "num_delay": moved}
                if self._total(inf.num_delay) > 0:
                    flag, k = self._repair()
                    if flag != 0:
                        return {"flag": flag, "num_neg": 0, "num_two": 0, "matrix_rank": 0,
                                "num_delay": moved}
                    moved += k
                    restart = True
                    break
                if phase == 1:
                    self._sum(self.xchg_factor, self.factor_elems)
            if not restart:
                break
        else:
            return {"flag": -98, "num_neg": 0, "num_two": 0, "matrix_rank": 0, "num_delay": moved}
        stats = self.torch.tensor([float(inf.num_neg), float(inf.num_two), float(self.n - inf.matrix_rank)],
                                  dtype=self.torch.float64)
        self._host_reduce(stats, self.dist.ReduceOp.SUM)
        warn = self._total(max(flags), self.dist.ReduceOp.MAX)
        return {"flag": int(warn), "num_neg": int(stats[0].item()), "num_two": int(stats[1].item()),
                "matrix_rank": self.n - int(stats[2].item()), "num_delay": moved,
                "num_factor": inf.num_factor, "num_flops": inf.num_flops, "nlevels": inf.nlevels,
                "num_sup": inf.num_sup}

    def _host_reduce(self, t, op):
        if self.direct:
            d = t.cuda()
            self.dist.all_reduce(d, op=op, group=self.group)
            t.copy_(d.cpu())
        else:
            self.dist.all_reduce(t, op=op, group=self.group)

    # ---- SLS_solve (one right-hand side, job "all") ----------------------------------------------------
    def solve_dev(self, d_x):
        """d_x: device tensor (n,) with the right-hand side on every rank; overwritten by the solution
        on every rank."""
        s = self.sls
        px, pb = C.c_void_p(d_x.data_ptr()), C.c_void_p(self.xchg_solve.data_ptr())

        def phase(k):
            inf = Inform()
            self.torch.cuda.current_stream().synchronize()
            flag = lib.gsls_shard_solve_dev(s.handle, k, px, pb, C.byref(inf))
            if flag < 0:
                raise RuntimeError("gsls_shard_solve_dev phase %d failed with flag %d" % (k, flag))

        phase(1)
        self._sum(self.xchg_solve, self.cvec_elems)
        phase(2)
        self._bcast0(self.xchg_solve, self.n)
        phase(3)
        self._sum(self.xchg_solve, self.n)
        phase(4)
        return d_x

    def partition(self):
        """(owner[nnodes], cutroots) -- owner -1 is the top part run by rank 0"""
        import numpy as np
        nn = self.sls.symbolic()["sptr"].shape[0] - 1
        owner = np.zeros(max(nn, 1), dtype=np.int32)
        ncut = C.c_int32()
        lib.gsls_shard_get(self.sls.handle, owner.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ncut), None)
        cut = np.zeros(max(ncut.value, 1), dtype=np.int32)
        lib.gsls_shard_get(self.sls.handle, None, None, cut.ctypes.data_as(C.POINTER(C.c_int32)))
        return owner[:nn], cut[: ncut.value] - 1
