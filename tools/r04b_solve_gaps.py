"""rocpd kernel trace of tools/r04b_solve_trace.py -> the LAST solve: kernel time by name, idle time, the gaps by what follows them."""
import sqlite3, sys
from collections import Counter, defaultdict
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
try:
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
except Exception:
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
# the last solve = from the last k_cam_cache-free stretch: find the last 4 big gaps (> 1 ms: host between solves)
cut = [i for i in range(1, len(rows)) if rows[i][1] - rows[i - 1][2] > 1_000_000]
rows = rows[cut[-1]:] if cut else rows
busy = sum(e - s for _, s, e in rows); span = rows[-1][2] - rows[0][1]
print("last solve: dispatches %d  span %.3f ms  busy %.3f ms (%.1f %%)  idle %.3f ms" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
dur = defaultdict(list)
for n, s, e in rows: dur[n.split("(")[0][:48]].append((e - s) / 1e3)
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])): print("  %-50s n %4d  avg %7.2f us  total %7.3f ms" % (n, len(v), sum(v) / len(v), sum(v) / 1e3))
gaps = [((rows[i + 1][1] - rows[i][2]) / 1e3, rows[i][0].split("(")[0][:28], rows[i + 1][0].split("(")[0][:28]) for i in range(len(rows) - 1)]
print("gaps: total %.3f ms; <2us %d, 2-10us %d, 10-30us %d, >30us %d" % (sum(g[0] for g in gaps) / 1e3, sum(g[0] < 2 for g in gaps), sum(2 <= g[0] < 10 for g in gaps), sum(10 <= g[0] < 30 for g in gaps), sum(g[0] >= 30 for g in gaps)))
big = [g for g in gaps if g[0] >= 10]
print("gaps >= 10 us: total %.3f ms, by (before -> after):" % (sum(g[0] for g in big) / 1e3))
for (a, b), n in Counter((g[1], g[2]) for g in big).most_common(14): print("   %4d x  %-28s -> %-28s  %.1f us avg" % (n, a, b, sum(g[0] for g in big if (g[1], g[2]) == (a, b)) / n))
