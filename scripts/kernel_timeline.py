#!/usr/bin/env python
"""last N kernel dispatches of a rocprofv3 rocpd database as a timeline: start (us, relative), duration, queue, name
usage: kernel_timeline.py <run_results.db> [N]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute(f"select start, end, {q or '0'}, {name_col} from kernels order by start"))
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = rows[-n - skip:len(rows) - skip] if len(rows) > n + skip else rows[-n:]
t0 = rows[0][0]
print("# columns:", cols)
for s, e, qq, nm in rows:
    print("%10.2f %8.2f q%-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qq, nm[:60]))
