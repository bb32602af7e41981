"""One sparse system factorized across several GPUs: elimination-tree sharding.

The reference deals independent subtrees of the assembly tree to NUMA regions / GPUs
(find_subtree_partition, src/ssids/anal.f90:284-459; assignment :569-590) and keeps the top of the
tree on one owner.  Here that is one process per GPU: every rank analyses the same matrix (integer work,
deterministic, identical on all ranks), `gsls_shard` deals the subtrees, and the only data that cross between ranks
are the CUT ROOTS' blocks and vectors (a cut root = the root of a dealt subtree):

  * factorize: their contribution blocks (+ 8 counters) REDUCED onto rank 0, 16 status words BROADCAST back;
  * solve:     their contribution vectors REDUCED onto rank 0, their z-vectors (the ancestors' part of the solution
               each subtree needs) BROADCAST back.  Every rank ends with the solution of the variables it eliminated.

No O(n) collective on the data path (`collect` is the optional all-reduce for callers that want the whole vector on
every rank).  Every reduced element is non-zero on exactly one rank, so results do not depend on the reduction order.

Two transports for the same protocol:
  * mode "lib":   the exchange happens INSIDE libgsls.so (RCCL on the handle's stream, gsls_comm_*): what a Fortran
                  host uses; this module only distributes the 128-byte communicator id;
  * mode "torch": this module issues the collectives on the exchange buffers with torch.distributed (RCCL when the
                  process group is "nccl"; staged through host memory with "gloo", which is what the tests on a
                  one-GPU box use).
"""
import ctypes as C
import os

from ._lib import Inform, lib


class TreeShardedSLS:
    """Wraps an analysed galahad_amd.SLS object; all ranks must call every method collectively."""

    def __init__(self, sls, group=None, d_val=None, mode=None):
        """d_val (device tensor with the matrix values, optional): lets every rank refine its own ordering with
        the values (gsls_refine_order_dev) before the tree is dealt -- same result on all ranks.
        mode: "lib" | "torch" (default: $GSLS_SHARD_COMM, else "torch")"""
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
        self.n = sls.n
        self.direct = dist.get_backend(group) == "nccl"
        self.mode = mode or os.environ.get("GSLS_SHARD_COMM", "torch")
        if self.mode == "lib":
            ident = [None]
            if self.rank == 0:
                buf = C.create_string_buffer(128)
                if lib.gsls_comm_unique_id(buf) != 0:
                    raise RuntimeError("gsls_comm_unique_id failed")
                ident[0] = buf.raw
            dist.broadcast_object_list(ident, src=self._src0(), group=group)
            flag = lib.gsls_comm_init(sls.handle, self.world, self.rank, ident[0], C.byref(sls.opts))
            if flag != 0:
                raise RuntimeError("gsls_comm_init failed with flag %d" % flag)
        else:
            ce, ve = C.c_int64(), C.c_int64()
            flag = lib.gsls_shard(sls.handle, self.world, self.rank, C.byref(ce), C.byref(ve))
            if flag != 0:
                raise RuntimeError("gsls_shard failed with flag %d" % flag)
            self._alloc(ce.value, ve.value)

    def _src0(self):
        return self.dist.get_global_rank(self.group, 0) if self.group is not None else 0

    def _cut_vector_elems(self):
        import numpy as np
        sym = self.sls.symbolic()
        _, cut = self.partition()
        return int((np.diff(sym["rptr"])[cut] - np.diff(sym["sptr"])[cut]).sum())

    # ---- collectives on a slice of a device buffer (mode "torch") ------------------------------------------
    def _coll(self, what, buf, lo, hi):
        if hi <= lo:
            return
        t = buf[lo:hi]
        self.torch.cuda.current_stream().synchronize()
        h = t if self.direct else t.cpu()
        if what == "reduce0":
            self.dist.reduce(h, dst=self._src0(), op=self.dist.ReduceOp.SUM, group=self.group)
        elif what == "bcast0":
            self.dist.broadcast(h, src=self._src0(), group=self.group)
        else:
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
        if not self.direct:
            t.copy_(h)
        self.torch.cuda.current_stream().synchronize()

    def _alloc(self, ce, ve):
        torch = self.torch
        self.factor_elems, self.solve_elems = ce, ve
        dev = torch.device("cuda", torch.cuda.current_device())
        self.xchg_factor = torch.zeros(ce, dtype=torch.float64, device=dev)
        self.xchg_solve = torch.zeros(ve, dtype=torch.float64, device=dev)
        self.cvec_elems = self._cut_vector_elems()

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

    # ---- SLS_factorize ---------------------------------------------------------------------------------
    def factorize_dev(self, d_val, posdef, max_pass=60):
        """d_val: device tensor holding VAL (SLS.scatter_values) on every rank.  Returns a dict with the
        flag and the statistics of the whole matrix (the same on every rank)."""
        s = self.sls
        if self.mode == "lib":
            inf = Inform()
            self.torch.cuda.current_stream().synchronize()
            flag = lib.gsls_comm_factor_dev(s.handle, 1 if posdef else 0, C.c_void_p(d_val.data_ptr()),
                                            C.byref(s.opts), C.byref(inf))
            return {"flag": int(flag), "num_neg": inf.num_neg, "num_two": inf.num_two, "matrix_rank": inf.matrix_rank,
                    "num_delay": inf.num_delay, "num_factor": inf.num_factor, "num_flops": inf.num_flops,
                    "nlevels": inf.nlevels, "num_sup": inf.num_sup}
        moved = 0
        for _ in range(max_pass + 1):
            E = self.factor_elems
            inf = Inform()
            bad = 0
            for phase in (1, 2):
                self.torch.cuda.current_stream().synchronize()
                flag = lib.gsls_shard_factor_dev(s.handle, phase, 1 if posdef else 0,
                                                 C.c_void_p(d_val.data_ptr()),
                                                 C.c_void_p(self.xchg_factor.data_ptr()), C.byref(s.opts),
                                                 C.byref(inf))
                if flag < 0 and flag not in (-5, -6):      # (singular / not positive definite travel in the status words)
                    bad = flag
                if phase == 1:
                    # the cut roots' blocks and this rank's counters in ONE reduction onto rank 0
                    self._coll("reduce0", self.xchg_factor, 0, E - 16)
            self._coll("bcast0", self.xchg_factor, E - 24, E - 8)
            st = self.xchg_factor[E - 24: E - 8].cpu().numpy()
            worst = self._host_min(bad)
            if worst < 0:
                return {"flag": int(worst), "num_neg": 0, "num_two": 0, "matrix_rank": 0, "num_delay": moved}
            tot = st[:8] + st[8:]
            if not posdef and tot[5] > 0:
                # a front on the wave-per-front path wanted pivoting: the same factorization again without that path
                lib.gsls_shard_fast(s.handle, 0)
                self._fast_off = True
                continue
            if posdef and tot[0] > 0:
                return {"flag": -6, "num_neg": 0, "num_two": 0, "matrix_rank": 0, "num_delay": moved}
            if not posdef and tot[1] > 0:
                lib.gsls_shard_fast(s.handle, 0)
                flag, k = self._repair()
                if flag != 0:
                    return {"flag": flag, "num_neg": 0, "num_two": 0, "matrix_rank": 0, "num_delay": moved}
                moved += k
                continue
            if not posdef:      # no block needed pivoting on any rank: the next factorization takes the wave kernels
                lib.gsls_shard_fast(s.handle, 1 if (tot[6] == 0 and not getattr(self, "_fast_off", False)) else 0)
            nzero = int(tot[4])
            flag = 0
            if nzero > 0:
                flag = 7 if s.opts.action else -5
            return {"flag": flag, "num_neg": int(tot[2]), "num_two": int(tot[3]), "matrix_rank": self.n - nzero,
                    "num_delay": moved, "num_factor": inf.num_factor, "num_flops": inf.num_flops,
                    "nlevels": inf.nlevels, "num_sup": inf.num_sup}
        return {"flag": -98, "num_neg": 0, "num_two": 0, "matrix_rank": 0, "num_delay": moved}

    def _host_min(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64)
        if self.direct:
            d = t.cuda()
            self.dist.all_reduce(d, op=self.dist.ReduceOp.MIN, group=self.group)
            return d.item()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return t.item()

    # ---- SLS_solve (one right-hand side, job "all") ----------------------------------------------------
    def solve_dev(self, d_x, collect=True):
        """d_x: device tensor (n,) with the right-hand side on every rank.  On return the entries of the variables
        this rank eliminated hold the solution (rank 0: also the top part's); with collect=True (an extra O(n)
        all-reduce, not part of the solve proper) every rank holds the whole solution."""
        s = self.sls
        px = C.c_void_p(d_x.data_ptr())
        if self.mode == "lib":
            inf = Inform()
            self.torch.cuda.current_stream().synchronize()
            flag = lib.gsls_comm_solve_dev(s.handle, px, C.byref(inf))
            if flag < 0:
                raise RuntimeError("gsls_comm_solve_dev failed with flag %d" % flag)
            if collect:
                flag = lib.gsls_comm_collect_dev(s.handle, px, C.byref(inf))
                if flag < 0:
                    raise RuntimeError("gsls_comm_collect_dev failed with flag %d" % flag)
            return d_x
        pb = C.c_void_p(self.xchg_solve.data_ptr())

        def phase(k):
            inf = Inform()
            self.torch.cuda.current_stream().synchronize()
            flag = lib.gsls_shard_solve_dev(s.handle, k, px, pb, C.byref(inf))
            if flag < 0:
                raise RuntimeError("gsls_shard_solve_dev phase %d failed with flag %d" % (k, flag))

        phase(1)
        self._coll("reduce0", self.xchg_solve, 0, self.cvec_elems)
        phase(2)
        self._coll("bcast0", self.xchg_solve, 0, self.cvec_elems)
        phase(3)
        if collect:
            phase(4)
            self._coll("sum", self.xchg_solve, 0, self.n)
            phase(5)
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
