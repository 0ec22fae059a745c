#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel stats + PMC counters) as text/CSV.

usage: rocpd_summary.py <run_results.db> [--pmc]
The kernel-stats table is the same information `rocprofv3 --stats` prints (per-kernel calls, total,
average, min, max in ns); the PMC table is the per-kernel mean of every collected counter."""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    print("# kernel stats (ns):", db)
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,pct")
    rows = list(cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    for r in rows:
        print(f"\"{r[0][:90]}\",{r[1]},{r[2]},{r[3]:.0f},{r[4]},{r[5]},{100.0*r[2]/tot:.2f}")
    if "--pmc" in sys.argv:
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        print("# counters_collection columns:", ccols)
        kn = "kernel_name" if "kernel_name" in ccols else name_col
        cn = "counter_name" if "counter_name" in ccols else "name"
        vn = "value" if "value" in ccols else "counter_value"
        print("kernel,counter,dispatches,mean_value,sum_value")
        for r in cur.execute(f"select {kn}, {cn}, count(*), avg({vn}), sum({vn}) from counters_collection group by {kn}, {cn} order by 1, 2"):
            print(f"\"{r[0][:90]}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]:.1f}")


if __name__ == "__main__":
    main()
