"""rocpd kernel trace of tools/r05_c4_trace.py -> the timeline of ONE LM iteration of the last solve (argument 2: which, counted in
k_comp_activity launches from the end of the trace; default 10): every dispatch with its start offset, duration and the gap in front of it
on its own queue -- where an iteration of the component path spends its time when only the slow scene is live."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute("select s.kernel_name, d.start, d.end, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start" % (qcol, kd, ks)).fetchall()
acts = [i for i, r in enumerate(rows) if "k_comp_activity" in r[0]]
i0, i1 = acts[-back - 1], acts[-back]
# an iteration as the host sees it: from the first kernel after the previous scatter's followers ... simply activity to activity
t0 = rows[i0][1]
print("LM iteration %d from the end: %.1f us from k_comp_activity to the next k_comp_activity, %d dispatches" % (back, (rows[i1][1] - t0) / 1e3, i1 - i0))
last_end = {}
agg = {}
for n, s, e, q in rows[i0:i1]:
    nm = n.split("(")[0].replace("void gsfm::", "").replace("gsfm::", "")[:34]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    a = agg.setdefault((q, nm), [0, 0.0, 0.0, (s - t0) / 1e3, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[2] += gap; a[4] = (e - t0) / 1e3
print("queue  kernel                              n   busy us   gaps us   first start   last end")
for (q, nm), a in sorted(agg.items(), key=lambda kv: kv[1][3]):
    print("%5s  %-34s %3d  %8.1f  %8.1f  %10.1f  %10.1f" % (q, nm, a[0], a[1], a[2], a[3], a[4]))
