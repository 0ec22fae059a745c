#!/usr/bin/env python
"""r12s: the busiest queue of a rocprofv3 rocpd kernel trace: how much of the second half of the run it was busy, and the idle gaps between its consecutive kernels
by the kernel that FOLLOWS the gap.  usage: queue_gaps.py <run_results.db>"""
import collections, sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
q = "queue_id" if "queue_id" in cols else "stream_id"
rows = list(cur.execute(f"select start, end, {q}, {name_col} from kernels order by start"))
t0, t1 = rows[0][0], rows[-1][1]
half = t0 + (t1 - t0) // 2
rows = [r for r in rows if r[0] >= half]
busy = collections.Counter()
for s, e, qq, n in rows: busy[qq] += e - s
main = busy.most_common(1)[0][0]
mine = [r for r in rows if r[2] == main]
span = mine[-1][1] - mine[0][0]
print("queue %s: %d kernels over %.1f ms, busy %.1f %%" % (main, len(mine), span / 1e6, 100.0 * busy[main] / span))
per = collections.defaultdict(list)
for a, b in zip(mine, mine[1:]):
    per[b[3][:50]].append((b[0] - a[1]) / 1e3)
for n, g in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    big = [x for x in g if x > 2.0]
    print("  before %-50s %5d gaps, total %8.1f us, mean %6.2f us; over 2 us: %4d, their mean %6.1f us" % (n, len(g), sum(g), sum(g) / len(g), len(big), sum(big) / len(big) if big else 0.0))
dur = collections.defaultdict(list)
for s, e, qq, n in mine: dur[n[:50]].append((e - s) / 1e3)
for n, d in dur.items(): print("  %-50s mean %7.2f us (%d)" % (n, sum(d) / len(d), len(d)))
others = [(r[3][:40], (r[1] - r[0]) / 1e3) for r in rows if r[2] != main]
oc = collections.defaultdict(list)
for n, d in others: oc[n].append(d)
for n, d in oc.items(): print("  other queues: %-40s %5d launches, mean %7.2f us" % (n, len(d), sum(d) / len(d)))
for n, d in dur.items():
    d = sorted(d)
    print("  %-50s percentiles 5 / 25 / 50 / 75 / 95 / 99: %s" % (n, " / ".join("%.1f" % d[min(len(d) - 1, int(len(d) * p / 100))] for p in (5, 25, 50, 75, 95, 99))))
