#!/usr/bin/env python
"""r12w: which hardware queues the two halves of bench.py's double-buffered closed loop ran on (rocprofv3 rocpd database): the single-tick step launches by
(queue, grid size), and for the half-sized ones how often consecutive launches of different queues overlapped in time.  usage: queues_of_halves.py <db>"""
import collections, sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
q = "queue_id" if "queue_id" in cols else "stream_id"
grid = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
rows = list(cur.execute(f"select start, end, {q}, {grid}, {name_col} from kernels order by start"))
c = collections.Counter((qq, g, n[:40]) for s, e, qq, g, n in rows if "step_kernel<1>" in n or "raster_fast_kernel" in n)
for k, v in sorted(c.items(), key=lambda kv: -kv[1]): print(v, k)
steps = [(s, e, qq, g) for s, e, qq, g, n in rows if "step_kernel<1>" in n]
gmin = min(g for _, _, _, g in steps)
half = [(s, e, qq) for s, e, qq, g in steps if g == gmin]
qs = sorted(set(qq for _, _, qq in half))
print("half-sized step launches: %d on queues %s" % (len(half), qs))
ov = sum(1 for a, b in zip(half, half[1:]) if a[2] != b[2] and b[0] < a[1])
print("consecutive half-sized step launches of different queues that overlapped: %d of %d" % (ov, len(half) - 1))
if half: print("span %.1f ms -> %.1f us per pair of half steps" % ((half[-1][1] - half[0][0]) / 1e6, (half[-1][1] - half[0][0]) / 1e3 / (len(half) / 2)))
