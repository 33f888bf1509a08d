#!/usr/bin/env python3
"""Per-kernel averages of every counter in one rocprofv3 --pmc pass (ROCm 7.2 rocpd SQLite output).
usage: summarize_pmc_db.py <results.db> [kernel-name substring ...]
Prints, per kernel (longest first): launches, mean duration, and for every counter the mean value per launch."""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    pats = sys.argv[2:]
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration), grid_size, workgroup_size, vgpr_count, sgpr_count, lds_block_size "
                     "from counters_collection group by kernel_name, counter_name").fetchall()
    by = {}
    for k, cn, n, v, dur, grid, wg, vg, sg, lds in rows:
        if pats and not any(p in k for p in pats):
            continue
        e = by.setdefault(k, {"n": n, "dur": dur, "grid": grid, "wg": wg, "vgpr": vg, "sgpr": sg, "lds": lds, "c": {}})
        e["c"][cn] = v
    for k in sorted(by, key=lambda k: -by[k]["dur"] * by[k]["n"]):
        e = by[k]
        if e["dur"] * e["n"] < 1e5:   # < 0.1 ms in total
            continue
        print("%s\n   launches %d, mean duration (profiled) %.1f us, grid %d x wg %d, vgpr %d sgpr %d lds %d" % (k[:150], e["n"], e["dur"] / 1e3, e["grid"], e["wg"], e["vgpr"], e["sgpr"], e["lds"]))
        for cn in sorted(e["c"]):
            print("   %-44s %18.1f" % (cn, e["c"][cn]))


if __name__ == "__main__":
    main()
