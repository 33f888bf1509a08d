#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (ROCm 7.2 rocpd SQLite output):
   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- <cmd> ; rocprofv3 --kernel-trace --pmc WRITE_SIZE -- <cmd>
usage: summarize_pmc.py <fetch.db> <write.db> [out.json]
FETCH_SIZE / WRITE_SIZE are KiB; fetch is doubled (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced read); WRITE_SIZE is taken as is (uncalibrated)."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection where counter_name = ? "
                     "group by kernel_name order by 3 desc", (counter,)).fetchall()
    return {r[0]: {"calls": r[1], "avg": r[2], "min": r[3], "max": r[4]} for r in rows}


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    print("# HBM traffic per launch, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); fetch_corrected = 2 x FETCH_SIZE x 1024 B, write = WRITE_SIZE x 1024 B")
    print("%-70s %6s %16s %20s %16s" % ("kernel", "calls", "FETCH_SIZE[KiB]", "fetch_corrected[MB]", "WRITE_SIZE[MB]"))
    out = {}
    for k in sorted(f, key=lambda k: -f[k]["avg"]):
        wk = w.get(k, {"avg": 0.0})
        fb, wb = 2.0 * f[k]["avg"] * 1024.0, wk["avg"] * 1024.0
        if fb + wb < 1e6:
            continue
        print("%-70s %6d %16.0f %20.1f %16.1f" % (k[:70], f[k]["calls"], f[k]["avg"], fb / 1e6, wb / 1e6))
        out[k] = {"fetch_bytes": int(fb), "write_bytes": int(wb), "calls": f[k]["calls"]}
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
