#!/usr/bin/env python3
"""bench.py -- SLS factorize+solve throughput of the gsls (MI355X) backend.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.
  * workload (N=1, default `--workload kkt`): the configuration BASELINE.json's metric is quoted on --
    "n=1e6 banded KKT": K = [H A^T; A 0], H = tridiag(2,-1)+diag(sigma) of order n=1e6, m=2e5 constraint
    rows (QPBAND pattern), order N=1.2e6, fp64 (configs[2] shape, generator tests/problems.py:kkt_qpband,
    seed 20240102), factorized as a pivoted LDL^T (pivot_control=1, u=0.01, node_amalgamation=24 for GPU and CPU
    runs alike) through the SLS C ABI.
    `--workload band` is configs[1] (banded SPD n=1e5, semi-bandwidth 127, Cholesky).
  * one step  = SLS_factorize + SLS_solve of that system (gsls_factor_dev + gsls_solve_dev): matrix
    values and right-hand side are resident in HBM when the clock starts.  Analyse (symbolic, host
    integer work) is outside the metric, as in the reference's own timers (inform%time%clock_factorize
    / clock_solve, src/sls/sls.f90:4676-4683, 4951-4958).  So is the FIRST factorization of an
    indefinite matrix: it learns which pivots must be delayed and repairs the static elimination order
    (DESIGN.md); the warm-up steps absorb it, every timed step is a refactorization as an
    interior-point iteration would issue it.
  * value     = N * K * F / t.  kkt: F = flops_elimination of the elimination order actually used,
    which the CPU baseline is given as PERM (same pattern, PERM, nemin => the reference reports the
    same F; tests pin that equality bit-exactly).  band: F = the reference's flops_elimination in the
    NATURAL order (2 065 810 544), whatever ordering the GPU run chose, so reordering cannot inflate it.
  * N > 1     = ONE system, the subtrees of its elimination tree dealt to the GPUs, the cut roots' contribution
    blocks / vectors reduced onto rank 0 (--shard tree, the default: strong scaling); --shard replicas: N independent
    systems, one per rank (weak scaling); time = max over ranks.
  * roofline  = triangular-solve sweep against HBM: algorithmic bytes of one solve
    (2*8*nnz(L) + 4*8*n [+ 16 n indefinite], SURVEY.md section 8d, nnz(L) of the ordering used)
    / HIP-event time of the sweep's kernels on the library's stream.
  * cpu_baseline = the real reference (oracle/_ref/ref_driver: GALAHAD SLS + SPRAL SSIDS CPU,
    vendored reference BLAS) on the same matrix with the same PERM; rank 0, N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
STREAM_CEILING_GBS = 6200.0  # what streaming loads measure on the part (same guide; tools/wsolve_ubench.hip: 5.6 - 6.2 TB/s)
F_NATURAL_CFG2 = 2065810544  # reference flops_elimination, cfg2, natural order


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["kkt", "band"], default="kkt",
                    help="kkt: BASELINE.json's metric config -- KKT saddle point n=1e6, m=2e5 (configs[2] shape), "
                         "pivoted LDL^T; band: configs[1], banded SPD n=1e5 semi-bandwidth 127, Cholesky")
    ap.add_argument("--n", type=int, default=0, help="kkt: primal dimension (default 1e6); band: order (default 1e5)")
    ap.add_argument("--m", type=int, default=0, help="kkt: number of constraints (default n/5)")
    ap.add_argument("--semibw", type=int, default=127)
    ap.add_argument("--ordering", choices=["free", "natural"], default="free",
                    help="free: the backend's own ordering (perf run); natural: identity PERM (parity run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-facade", action="store_true", help="skip the SLS / SBLS facade timings (extra block `facade`)")
    ap.add_argument("--nemin", type=int, default=0, help="supernode amalgamation (0: backend default)")
    ap.add_argument("--shard", choices=["replicas", "tree"], default="tree",
                    help="N>1: tree (default) = ONE system, elimination-tree subtrees dealt to the GPUs, the cut roots' "
                         "blocks reduced onto rank 0 (the north-star's sharded metric: strong scaling); replicas = one "
                         "independent system per GPU (weak scaling, a solver farm)")
    ap.add_argument("--drift", type=int, default=0,
                    help="kkt only: after the timed steps, run this many extra steps in which the diagonal of H changes "
                         "like barrier terms of an interior-point loop (x 10^U(0,2) per entry and step: H stays positive "
                         "definite, the inertia (n, m, 0) must not change) and report "
                         "their time and how many blocks needed the pivoting fallback (extra field `drift`)")
    ap.add_argument("--drift-decades", type=float, default=2.0, help="width of the drift: x 10^U(0, this)")
    ap.add_argument("--one-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (with --backend gloo); never a measurement")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    a = ap.parse_args()
    if a.n <= 0:
        a.n = 1000000 if a.workload == "kkt" else 100000
    if a.m <= 0:
        a.m = a.n // 5
    return a


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(prob, posdef, perm, nemin):
    """reference Fortran CPU path on the same matrix with the same PERM and nemin, bounded: a few
    repeats at two thread counts (the reference's OpenMP task code degrades badly when oversubscribed,
    so the best of {8, min(cores,16)} threads is reported -- the count used is stated in `cores`)"""
    from oracle import refio
    if not refio.available():
        return None
    n, row, col, val, rhs, xs = prob
    best = None
    sweep = []
    for threads in sorted({min(t, host_cores()) for t in (8, 16, 32, 64, 128)}):
        r = refio.run(n, row, col, val, rhs, perm=perm, pivot_control=2 if posdef else 1, nemin=nemin,
                      repeat=3, threads=threads, timeout=1500)
        if r["status_factorize"] != 0 or r["status_solve"] != 0:
            continue
        t = r["t_factorize_median"] + r["t_solve_median"]
        sweep.append((threads, round(r["flops_elimination"] / t / 1e9, 2)))
        if best is None or t < best[0]:
            best = (t, threads, r)
    if best is None:
        return None
    t, threads, r = best
    # second baseline: the same reference build with the optimised OpenBLAS that ships inside
    # scipy behind SSIDS' BLAS / LAPACK calls (oracle/blas_shim.c, LD_PRELOAD), at the best thread count found above
    ob = None
    if refio.openblas_available():
        try:
            r2 = refio.run(n, row, col, val, rhs, perm=perm, pivot_control=2 if posdef else 1, nemin=nemin,
                           repeat=3, threads=threads, timeout=1500, blas="openblas")
            if r2["status_factorize"] == 0 and r2["status_solve"] == 0:
                t2 = r2["t_factorize_median"] + r2["t_solve_median"]
                ob = {"value": r2["flops_elimination"] / t2 / 1e9, "factorize_s": r2["t_factorize_median"],
                      "solve_s": r2["t_solve_median"], "max_err": float(np.abs(r2["x"] - xs).max())}
        except Exception as e:
            ob = {"error": repr(e)[:200]}
    vendored = r["flops_elimination"] / t / 1e9
    value = max(vendored, ob["value"]) if ob and "value" in ob else vendored
    return {"value": value, "unit": "GF/s", "cores": threads,
            "kind": "reference",
            "sample": "full workload, same PERM and nemin as the GPU run, median of 3 SLS_factorize+SLS_solve "
                      "(ssids, OMP_NUM_THREADS=%d of %d usable cores); `value` is the better of two builds of the "
                      "reference: vendored reference BLAS %.2f GF/s (factorize %.3fs solve %.3fs analyse %.2fs), "
                      "OpenBLAS from scipy.libs behind SSIDS' BLAS/LAPACK calls %s; vendored-BLAS GF/s by OpenMP "
                      "thread count (one socket and beyond): %s"
                      % (threads, host_cores(), vendored, r["t_factorize_median"], r["t_solve_median"], r["t_analyse"],
                         ("%.2f GF/s" % ob["value"]) if ob and "value" in ob else "unavailable",
                         ", ".join("%d: %.2f" % tv for tv in sweep)),
            "vendored_blas": vendored, "openblas": ob,
            "thread_sweep": sweep,
            "flops_elimination": r["flops_elimination"], "entries_in_factors": r["entries_in_factors"],
            "delayed_pivots": r["delayed"], "negative_eigenvalues": r["negative_eigenvalues"],
            "max_err": float(np.abs(r["x"] - xs).max())}


def facade_timings(prob, a, nemin):
    """What a GALAHAD caller gets: the REAL SLS / SBLS facades (reference Fortran, built by oracle/build_ref.sh with the
    gsls arms of INTEGRATION.md) over this backend on the same system -- host arrays in, host x out, the facade's own
    clocks (inform%time%clock_factorize / clock_solve semantics: median over repeats, analyse excluded)."""
    from oracle import refio
    if not refio.dropin_available():
        return None
    n, row, col, val, rhs, xs = prob
    out = {}
    try:
        r = refio.run(n, row, col, val, rhs, solver="gsls", pivot_control=1, nemin=nemin, repeat=7, max_refine=0)
        r1 = refio.run(n, row, col, val, rhs, solver="gsls", pivot_control=1, nemin=nemin, repeat=7, max_refine=1)
        out["sls"] = {"gflops": r["flops_elimination"] / (r["t_factorize_median"] + r["t_solve_median"]) / 1e9,
                      "factorize_ms": r["t_factorize_median"] * 1e3, "solve_ms": r["t_solve_median"] * 1e3,
                      "solve_with_one_refinement_ms": r1["t_solve_median"] * 1e3,
                      "status": [r["status_analyse"], r["status_factorize"], r["status_solve"]],
                      "max_err": float(np.abs(r["x"] - xs).max())}
        if refio.sbls_available(dropin=True):
            i = np.arange(a.n)
            H = (np.concatenate([i, i[1:]]) + 1, np.concatenate([i, i[:-1]]) + 1, val[: 2 * a.n - 1])
            A = (np.concatenate([np.arange(a.m), np.arange(a.m)]) + 1,
                 np.concatenate([np.arange(a.m), a.m + np.arange(a.m)]) + 1, np.ones(2 * a.m))
            Cm = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0))
            rs = refio.run_sbls(a.n, a.m, H, A, Cm, rhs, solver="gsls", factorization=2, repeat=7, itref_max=1)
            out["sbls"] = {"gflops": r["flops_elimination"] / (rs["t_factorize_median"] + rs["t_solve_median"]) / 1e9,
                           "form_and_factorize_ms": rs["t_factorize_median"] * 1e3, "solve_ms": rs["t_solve_median"] * 1e3,
                           "status": [rs["status_factorize"], rs["status_solve"]],
                           "max_err": float(np.abs(rs["sol"] - xs).max()),
                           "note": "SBLS_solve_explicit (src/sbls/sbls.f90:5073-5388) with its refinement loop handed to the backend "
                                   "(integration/patch_sbls.py); on refactorizations A%val, H%val, -C%val go to the "
                                   "backend from the caller's arrays and K's values are put together in HBM "
                                   "(gsls_set_value_part), and the solve runs in SOL"}
    except Exception as e:      # the facade run is a report, not the metric
        out["error"] = repr(e)[:200]
    return out


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: this process starts the N ranks itself, as CHILD processes with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, and never touches the GPU or torch (no re-exec of a process that
    has initialised HIP).  Rank 0 prints the JSON line on the inherited stdout; a failing rank ends the run non-zero."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r if r > 0 else 1
                for q in live:          # the others would wait for it in a collective forever
                    q.terminate()
        time.sleep(0.05)
    sys.exit(rc)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if a.one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is what runs and what n_gpus reports"
              % (a.gpus, world), file=sys.stderr)
    if os.environ.get("GSLS_BENCH_SPAWN_ONLY"):     # (tests/test_dist.py: the rank plumbing without a GPU)
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world,
                          "master": os.environ.get("MASTER_ADDR", "") + ":" + os.environ.get("MASTER_PORT", "")}), flush=True)
        return
    import torch
    import torch.distributed as dist
    from galahad_amd import dist as gdist
    torch.cuda.set_device(local_rank)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import problems as P
    from galahad_amd import SLS, SMT, Control, InformSLS
    from galahad_amd._lib import Inform, lib

    # every rank owns one independent system of the same shape (different seed): weak scaling
    tree = a.shard == "tree" and world > 1
    seed_shift = 0 if tree else rank
    kkt = a.workload == "kkt"
    if kkt:
        prob = P.kkt_qpband(a.n, a.m, seed=20240102 + seed_shift)
        posdef = False
        name = ("KKT saddle point K=[H A^T; A 0]: H tridiag(2,-1)+diag(sigma) n=%d, m=%d QPBAND constraint rows, "
                "order %d, fp64, pivoted LDL^T (BASELINE.json metric config, configs[2] shape)" % (a.n, a.m, a.n + a.m))
    else:
        prob = P.banded_spd(a.n, a.semibw, seed=20240101 + seed_shift)
        posdef = True
        name = ("SLS standalone: random banded SPD n=%d, semi-bandwidth=%d, fp64 (BASELINE.json configs[1])"
                % (a.n, a.semibw))
    n, row, col, val, rhs, xs = prob
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, c, inf = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, inf)
    c.pivot_control = 2 if posdef else 1
    s.opts.device = local_rank
    perm = np.arange(1, n + 1) if a.ordering == "natural" else None
    if a.nemin > 0:
        c.node_amalgamation = a.nemin
    elif kkt:
        c.node_amalgamation = 24    # this tree is tens of thousands of tiny fronts: wider supernodes only add explicit
        #                             zeros (the reference's default is 32, src/ssids/datatypes.f90:222-245; 24 keeps 95 %
        #                             of the fronts within one wavefront: n <= 32, m <= 64).  The CPU baseline gets the same.
    t0 = time.perf_counter()
    s.analyse(m, c, inf, PERM=perm)
    t_analyse = time.perf_counter() - t0
    assert inf.status == 0, inf.status
    s._copy_control(c)

    # inputs resident in HBM before the clock starts
    VAL = s.scatter_values(m)
    d_val = torch.from_numpy(VAL).cuda()
    d_rhs = torch.from_numpy(rhs).cuda()
    d_x = torch.empty_like(d_rhs)
    ginf = Inform()

    tsh = None
    if tree:
        from galahad_amd.shard import TreeShardedSLS
        tsh = TreeShardedSLS(s, d_val=None if posdef else d_val)

    def step():
        if tsh is not None:
            st = tsh.factorize_dev(d_val, posdef)
            assert st["flag"] >= 0, st
            ginf.num_factor, ginf.num_flops = st["num_factor"], st["num_flops"]
            ginf.nlevels, ginf.num_sup = st["nlevels"], st["num_sup"]
            ginf.num_neg, ginf.num_two = st["num_neg"], st["num_two"]
            ginf.num_delay = max(ginf.num_delay, st["num_delay"])
            d_x.copy_(d_rhs)
            tsh.solve_dev(d_x, collect=False)     # the solve proper: cut vectors up, z-vectors down, no O(n) collective
            return
        f = lib.gsls_factor_dev(s.handle, 1 if posdef else 0, C.c_void_p(d_val.data_ptr()), None, C.byref(s.opts),
                                C.byref(ginf))
        assert f >= 0, f
        # right-hand side in d_rhs (kept), solution into d_x: gsls_solve_dev_rhs reads the right-hand side where it is (the
        # LDL^T wave tier gathers it anyway) or copies it on the handle's own stream -- no copy launched from here, no host
        # synchronisation between the factorization and the solve
        f = lib.gsls_solve_dev_rhs(s.handle, 0, 1, C.c_void_p(d_rhs.data_ptr()), C.c_void_p(d_x.data_ptr()), n,
                                   C.byref(s.opts), C.byref(ginf))
        assert f >= 0, f

    # the first factorization of an indefinite matrix repairs the elimination order (one-off, like analyse)
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    moved = ginf.num_delay
    for _ in range(max(a.warmup - 1, 0)):
        step()
    gdist.barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ksolve = []

    def sweep_seconds():
        kf, kd, kb = C.c_double(), C.c_double(), C.c_double()
        lib.gsls_last_solve_kernel_seconds(s.handle, C.byref(kf), C.byref(kd), C.byref(kb))
        return kf.value + kd.value + kb.value
    # gsls_solve_dev_rhs only enqueues: a step's sweep time (HIP events on the handle's stream) is read one step later,
    # after that step's factorization has synchronised the stream anyway -- no extra host round trip inside the loop
    for k in range(a.steps):
        if tsh is None and k > 0:
            fs = lib.gsls_factor_dev(s.handle, 1 if posdef else 0, C.c_void_p(d_val.data_ptr()), None, C.byref(s.opts),
                                     C.byref(ginf))
            assert fs >= 0, fs
            ksolve.append(sweep_seconds())
            fs = lib.gsls_solve_dev_rhs(s.handle, 0, 1, C.c_void_p(d_rhs.data_ptr()), C.c_void_p(d_x.data_ptr()), n,
                                        C.byref(s.opts), C.byref(ginf))
            assert fs >= 0, fs
        else:
            step()
            if tsh is not None:
                ksolve.append(sweep_seconds())
    torch.cuda.synchronize()
    if tsh is None:
        ksolve.append(sweep_seconds())
    gdist.barrier(world)
    elapsed = gdist.max_over_ranks(time.perf_counter() - t0, world)

    # correctness of what was timed (no refinement: SURVEY.md section 8d bars)
    if tsh is not None:       # every rank holds its own part of the solution: gather it once, outside the clock
        d_x.copy_(d_rhs)
        tsh.solve_dev(d_x, collect=True)
    x = d_x.cpu().numpy()
    res = P.scaled_residual(n, row, col, val, x, rhs)
    assert res <= (1e-13 if posdef else 1e-10), res

    drift = None
    if kkt and a.drift > 0 and tsh is None:
        # interior-point-like refactorizations: the diagonal of H (first entry of each of its CSC columns) drifts
        didx = torch.from_numpy((s.PTR[: a.n] - 1).astype(np.int64)).cuda()
        base = d_val[didx].clone()
        gen = torch.Generator(device="cuda")
        gen.manual_seed(7)
        times, piv = [], []
        fb, pb, bl = C.c_int32(), C.c_int32(), C.c_int32()
        for _ in range(a.drift):
            d_val[didx] = base * (10.0 ** (a.drift_decades * torch.rand(a.n, generator=gen, device="cuda", dtype=torch.float64)))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t1)
            lib.gsls_get_factor_stats(s.handle, C.byref(fb), C.byref(pb), C.byref(bl))
            piv.append(pb.value)
        d_val[didx] = base
        drift = {"steps": a.drift, "decades": a.drift_decades, "ms_per_step_median": float(np.median(times) * 1e3),
                 "ms_per_step_max": float(np.max(times) * 1e3), "pivoted_blocks_max": int(max(piv)),
                 "fast_blocks": fb.value, "tiny_blacklist": bl.value, "negative_eigenvalues": ginf.num_neg}

    if rank == 0:
        nnzL, flops_used = ginf.num_factor, ginf.num_flops      # of the order the timed steps used
        order = np.zeros(n, dtype=np.int32)
        lib.gsls_get_order(s.handle, order.ctypes.data_as(C.POINTER(C.c_int32)))
        if kkt or a.ordering == "natural":
            F = flops_used
        elif a.n == 100000 and a.semibw == 127:
            F = F_NATURAL_CFG2
        else:   # non-default band shape: natural-order flops from a second symbolic analyse
            s2, c2, i2 = SLS(), Control(), InformSLS()
            s2.initialize("gsls", c2, i2)
            s2.analyse(m, c2, i2, PERM=np.arange(1, n + 1))
            F = i2.flops_elimination
            s2.terminate()
        value = (1 if tree else world) * a.steps * F / elapsed / 1e9
        solve_bytes = 2 * 8 * nnzL + 4 * 8 * n + (0 if posdef else 16 * n)
        t_sweep = float(np.mean(ksolve))
        if tree or t_sweep <= 0:   # the sharded solve is four phases with collectives between: no single sweep
            t_sweep, achieved = None, None
        else:
            achieved = solve_bytes / t_sweep / 1e9
        # HBM bytes of one solve sweep from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE),
        # collected separately with rocprofv3 --pmc and committed; only valid for the profiled config
        traffic = None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic_%s.json" % a.workload)))
        pmc = cands[-1] if cands else ""      # the newest round's PMC pass (tools/measure_round.sh)
        default_shape = (kkt and a.n == 1000000 and a.m == 200000) or (not kkt and a.n == 100000 and a.semibw == 127)
        if pmc and os.path.exists(pmc) and default_shape and a.ordering == "free" and a.nemin == 0:
            with open(pmc) as f:
                traffic = json.load(f)["solve_sweep"]["hbm_bytes_corrected"]
        out = {
            "metric": "SLS factorize+solve GF/s (fp64) on n=1e6 banded KKT, 1/2/4/8 MI355X" if kkt
                      else "SLS factorize+solve GF/s (fp64)",
            "value": value, "unit": "GF/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if tree else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": name + ("; one system, tree-sharded over the GPUs" if tree else "; one system per GPU"),
                       "ordering": a.ordering, "node_amalgamation": c.node_amalgamation, "flops_numerator": F,
                       "flops_executed_per_step": flops_used, "entries_in_factors": nnzL,
                       "levels": ginf.nlevels, "supernodes": ginf.num_sup,
                       "negative_eigenvalues": ginf.num_neg, "two_by_two_pivots": ginf.num_two,
                       "pivots_moved_by_first_factorization": moved, "first_factorization_s": t_first,
                       "analyse_s": t_analyse, "scaled_residual": res},
            "roofline": {"bound": "hbm", "kernel": "triangular solve sweep (fwd+diag+bwd kernels)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if achieved is None else achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_launch": solve_bytes, "seconds_per_launch": t_sweep,
                         # `peak` is the vendor figure; plain 16-byte streaming loads reach 6.2 TB/s on this part
                         # (MI355X_MICROARCH.md, profiles/r02/calib and the round-3 micro-benchmark): the fraction of THAT
                         "frac_of_measured_stream_ceiling": None if achieved is None else achieved / STREAM_CEILING_GBS},
        }
        if drift is not None:
            out["drift"] = drift
        if world == 1 and kkt and not a.no_facade:
            fc = facade_timings(prob, a, c.node_amalgamation)
            if fc:
                out["facade"] = fc
        if world == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline(prob, posdef, order if (kkt or a.ordering == "natural") else np.arange(1, n + 1),
                              c.node_amalgamation)
            if cb is not None:
                out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    s.terminate()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
