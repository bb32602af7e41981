"""Aggregate a pmc_sq_*.txt table (tools/pmc_sum.py output of the SQ counter pass) by kernel: MFMA utilisation evidence
(SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CYCLES) as north_star / SURVEY 8(d) ask for."""
import sys, re, collections
rows = open(sys.argv[1]).read().splitlines()
hdr = rows[0].split()
cols = hdr[2:]
agg = collections.OrderedDict()
for ln in rows[1:]:
    m = re.match(r"^(.*?)\s+(\d+)\s+((?:[\d.e+]+\s*)+)$", ln)
    if not m:
        continue
    name = re.sub(r"\(.*", "", m.group(1)).strip()
    vals = [float(v) for v in m.group(3).split()]
    a = agg.setdefault(name, [0] + [0.0] * len(cols))
    a[0] += 1
    for i, v in enumerate(vals):
        a[1 + i] += v
ci = {c: i for i, c in enumerate(cols)}
def col(a, key):
    for c, i in ci.items():
        if key in c:
            return a[1 + i]
    return 0.0
print("%-44s %6s %14s %14s %14s %12s %10s %10s" % ("kernel (all launches of the profiled steps)", "calls", "MFMA_MOPS_F64", "MFMA_BUSY_CYC", "SQ_BUSY_CYC", "INSTS_VALU", "mfma/busy", "wait/wave"))
for name, a in agg.items():
    if not name.startswith(("gsls", "void gsls")):
        continue
    busy = col(a, "SQ_BUSY_CYCLES")
    print("%-44s %6d %14.4g %14.4g %14.4g %12.4g %10.3f %10.3f" % (name[-44:], a[0], col(a, "MFMA_MOPS"), col(a, "MFMA_BUSY"), busy, col(a, "INSTS_VALU"),
          col(a, "MFMA_BUSY") / busy if busy else 0.0, col(a, "SQ_WAIT_ANY") / col(a, "SQ_WAVE_CYCLES") if col(a, "SQ_WAVE_CYCLES") else 0.0))
