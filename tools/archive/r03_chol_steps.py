import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x from kernels where name like '%k_chol_step%' order by start").fetchall() if True else []
# last complete factorisation: 37 consecutive launches
rows = rows[-37:]
prev_end = None
for k, (n, s, e, g) in enumerate(rows):
    print("k %2d grid %6s dur %6.2f us gap_before %6.2f us" % (k, g, (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3))
    prev_end = e
