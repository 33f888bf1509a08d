#!/usr/bin/env python3
"""Turns a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace into the per-kernel stats table
committed under profiles/.  usage: summarize_rocpd.py <results.db> [title]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                     "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# %s" % title)
    print("# rocprofv3 --kernel-trace --stats ; times from the dispatch start/end timestamps")
    print("%-62s %7s %10s %10s %10s %10s %6s %5s %5s %6s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "scratch"))
    for r in rows:
        print("%-62s %7d %10.3f %10.2f %10.2f %10.2f %6.2f %5d %5d %6d %7d" % (r[0][:62], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6], r[7], r[8], r[9]))
    print("%-62s %7s %10.3f" % ("TOTAL", "", tot))


if __name__ == "__main__":
    main()
