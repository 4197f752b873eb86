"""Development aid: timeline of the LAST proof in a rocprofv3 (rocpd sqlite) kernel trace: busy time, idle gaps and
the kernels around the largest gaps.  usage: [LH_TRACE_COLUMNS=4 | LH_TRACE_SPLIT_IDLE_MS=50] python tools/trace_gaps.py <results.db> [kernel-name regex for a per-launch listing]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, workgroup_x, vgpr_count from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^(void )?(lh::|rocprim::\w+::detail::)?", "", n))[:44]
# the last proof starts at the last lasso iota kernel before which there is a long idle period
# (the access counters of a proof launch lasso_run_start_kernel once per chunk column; the proof begins with the sort in
# front of the first of them: walk back over the kernels that follow each other closely)
import os
split_ms = float(os.environ.get("LH_TRACE_SPLIT_IDLE_MS", "0"))  # > 0: the trace's tail after the last idle gap this long
if split_ms > 0:
    i0 = 0
    for i in range(1, len(rows)):
        if rows[i][1] - rows[i - 1][2] >= split_ms * 1e6:
            i0 = i
else:
    starts = [i for i, r in enumerate(rows) if "lasso_run_start" in r[0]]
    per = int(os.environ.get("LH_TRACE_COLUMNS", "1"))  # run-start launches per proof: 1 below 2^22 lookups (all columns together), else the chunk columns (range: 2, AND / XOR: 4)
    i0 = starts[-per]
    while i0 > 0 and rows[i0][1] - rows[i0 - 1][2] < 150e3:
        i0 -= 1
ks = rows[i0:]
t0, t1 = ks[0][1], max(r[2] for r in ks)
busy = sum(r[2] - r[1] for r in ks)
print("last proof: %d kernels, span %.2f ms, busy %.2f ms (%.0f%%)" % (len(ks), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
gaps = []
for a, b in zip(ks, ks[1:]):
    gaps.append((b[1] - a[2], short(a[0]), short(b[0]), (a[2] - t0) / 1e6))
print("gap histogram (us): ", end="")
for lo, hi in ((0, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1e9)):
    sel = [g[0] for g in gaps if lo * 1e3 <= g[0] < hi * 1e3]
    print("[%g,%g): %d = %.2f ms  " % (lo, hi, len(sel), sum(sel) / 1e6), end="")
print()
print("largest gaps:")
for g in sorted(gaps, reverse=True)[:18]:
    print("  %7.1f us at %6.2f ms  after %-44s before %s" % (g[0] / 1e3, g[3], g[1], g[2]))
agg = {}
for r in ks:
    a = agg.setdefault(short(r[0]), [0, 0])
    a[0] += 1
    a[1] += r[2] - r[1]
print("kernels of the proof:")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %-46s %4d  %8.3f ms  avg %7.1f us" % (n, c, d / 1e6, d / c / 1e3))
if len(sys.argv) > 2:  # detail: every launch of the kernels matching the pattern, by grid size
    pat = re.compile(sys.argv[2])
    print("launches matching %r (grid workgroups, duration us, gap before us):" % sys.argv[2])
    prev_end = None
    for r in ks:
        if pat.search(r[0]):
            print("  %-40s grid %6d x %4d  %7.1f us  gap %6.1f   [%8.3f .. %8.3f ms]" % (short(r[0]), r[3] // max(r[4], 1), r[4], (r[2] - r[1]) / 1e3,
                                                                 (r[1] - prev_end) / 1e3 if prev_end else 0.0, (r[1] - t0) / 1e6, (r[2] - t0) / 1e6))
        prev_end = r[2]
