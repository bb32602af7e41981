"""Sum the FETCH_SIZE / WRITE_SIZE counters (two rocprofv3 --pmc passes) over the kernels of the LAST
bench step: the solve sweep (k_permute_in .. k_permute_out) and the factorization (everything after the previous
step's solve kernels .. before k_permute_in).  FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide coalesced reads at
64 B, MI355X_MICROARCH.md); both counters are in KB."""
import csv, glob, json, sys


def load(d):
    import os
    f = max(glob.glob(d + '/*/*counter_collection.csv'), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    return rows


def split(rows):
    """last step: factorization = k_iota .. before k_permute_in; solve sweep = k_permute_in .. the last solve kernel
    (k_permute_out, or the wave tier's last backward launch, which writes the caller's vector itself)"""
    names = [r['Kernel_Name'] for r in rows]

    def is_solve(n):
        return any(t in n for t in ('k_wsolve_', 'k_solve_', 'k_permute_out', 'k_big_', 'k_permute_in'))
    pout = max(i for i, n in enumerate(names) if is_solve(n))
    pin = pout
    while pin > 0 and is_solve(names[pin - 1]):     # (whole solves read the right-hand side themselves: no k_permute_in)
        pin -= 1
    # the factorization starts right after the solve kernels of the step before (k_iota, the old marker, is not launched
    # by a refactorization that runs on the wave-per-front kernels only)
    a = pin - 1
    while a >= 0 and not is_solve(names[a]):
        a -= 1
    return rows[a + 1:pin], rows[pin:pout + 1]


def total(rows):
    return sum(float(r['Counter_Value']) for r in rows)


fr, wr = load(sys.argv[1]), load(sys.argv[2])
ff, fs = split(fr)
wf, ws = split(wr)
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, each with --kernel-trace only), "
               "bench.py --steps 2 --warmup 1 (%s); last step only; FETCH_SIZE doubled per MI355X_MICROARCH.md "
               "(gfx950 tallies 128-B requests of wide coalesced reads at 64 B); units KB*1024" % sys.argv[3],
       "solve_sweep": {"fetch_kb_raw": total(fs), "write_kb": total(ws), "launches": len(fs),
                       "hbm_bytes_corrected": int((2 * total(fs) + total(ws)) * 1024)},
       "factorize": {"fetch_kb_raw": total(ff), "write_kb": total(wf), "launches": len(ff),
                     "hbm_bytes_corrected": int((2 * total(ff) + total(wf)) * 1024)}}
print(json.dumps(out, indent=1))
