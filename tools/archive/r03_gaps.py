"""Kernel trace (rocpd db) -> busy time, gap time and the biggest gaps between consecutive dispatches of the last third of the run."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
rows = rows[len(rows) * 2 // 3:]
busy = sum(e - s for _, s, e in rows); span = rows[-1][2] - rows[0][1]
print("dispatches %d  span %.2f ms  busy %.2f ms (%.1f %%)" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span))
from collections import Counter, defaultdict
dur = defaultdict(list)
for n, s, e in rows: dur[n.split("(")[0][:40]].append((e - s) / 1e3)
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]: print("  %-42s n %5d  avg %6.2f us  total %7.2f ms" % (n, len(v), sum(v) / len(v), sum(v) / 1e3))
gaps = sorted(((rows[i + 1][1] - rows[i][2]) / 1e3, rows[i][0].split("(")[0][:30], rows[i + 1][0].split("(")[0][:30]) for i in range(len(rows) - 1))
big = [g for g in gaps if g[0] > 3.0]
print("gaps > 3 us: %d, total %.2f ms; histogram of those by what follows:" % (len(big), sum(g[0] for g in big) / 1e3), Counter(g[2] for g in big).most_common(6))
print("largest:", gaps[-5:])
