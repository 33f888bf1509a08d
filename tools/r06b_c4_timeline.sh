# C4 under rocprofv3 --kernel-trace: per-LM-iteration spans (k_comp_activity to k_comp_activity) of the last solve and the timeline of an early, a middle and a late iteration
cd /root/repo; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r06b_c4_timeline; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace -o tr -- python /root/repo/tools/r05_c4_trace.py > $OUT/c4.log 2>&1)
DB=$(find $OUT/trace -name '*.db' | head -1)
python - $DB > $OUT/spans.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute("select s.kernel_name, d.start, d.end, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start" % (qcol, kd, ks)).fetchall()
acts = [i for i, r in enumerate(rows) if "k_comp_activity" in r[0]][-46:]
print("queues seen:", sorted(set(r[3] for r in rows)))
for n, (a, b) in enumerate(zip(acts, acts[1:] + [len(rows)])):
    seg = rows[a:b]
    span = (seg[-1][2] - seg[0][1]) / 1e3
    chol = [r for r in seg if "k_chol" in r[0]]; pcg = [r for r in seg if "k_matvec_cg" in r[0] or "k_cg2" in r[0]]
    cb = sum(r[2] - r[1] for r in chol) / 1e3; pb = sum(r[2] - r[1] for r in pcg) / 1e3
    cspan = (chol[-1][2] - chol[0][1]) / 1e3 if chol else 0.0
    pspan = (pcg[-1][2] - pcg[0][1]) / 1e3 if pcg else 0.0
    # overlap: time during which kernels of both queues run
    print("it %2d: span %7.1f us  dispatches %4d  chol n %3d busy %6.1f span %6.1f | pcg n %3d busy %6.1f span %6.1f | other busy %6.1f" % (n + 1, span, len(seg), len(chol), cb, cspan, len(pcg), pb, pspan, sum(r[2] - r[1] for r in seg) / 1e3 - cb - pb))
PY
for b in 44 30 10 2; do python tools/r05_c4_timeline.py $DB $b > $OUT/timeline_back$b.txt 2>&1; done
find $OUT -name '*.db' -delete
tail -3 $OUT/c4.log; cat $OUT/spans.txt; cat $OUT/timeline_back10.txt
