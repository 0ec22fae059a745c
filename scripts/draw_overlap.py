#!/usr/bin/env python
"""r12h: from a rocprofv3 rocpd database of a device-fed Collect run: the observation launches' durations with and without a collect_draw_kernel launch
running beside them, the draw launches' durations and grid sizes, over the second half of the run.  usage: draw_overlap.py <run_results.db>"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
grid = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
rows = list(cur.execute(f"select start, end, {name_col}, {grid or '0'} from kernels order by start"))
t0, t1 = rows[0][0], rows[-1][1]
half = t0 + (t1 - t0) // 2
draws = [(s, e, g) for s, e, n, g in rows if "collect_draw" in n and s >= half]
copies = [(s, e) for s, e, n, g in rows if "blob_copy" in n and s >= half]
ras = [(s, e) for s, e, n, g in rows if "raster_glist_batch" in n and s >= half]
steps = [(s, e) for s, e, n, g in rows if "step_collect_ticks" in n and s >= half]
def overlap(s, e):
    return sum(max(0, min(e, de) - max(s, ds)) for ds, de, _ in draws)
w = [(e - s) / 1e3 for s, e in ras if overlap(s, e) > 0.5 * (e - s)]
wo = [(e - s) / 1e3 for s, e in ras if overlap(s, e) == 0]
mean = lambda v: sum(v) / len(v) if v else float("nan")
print("second half of the run: %.1f ms" % ((t1 - half) / 1e6))
print("draw launches: %d, mean %.0f us, max %.0f us, mean grid %s, busy %.0f %% of the time" % (len(draws), mean([(e - s) / 1e3 for s, e, _ in draws]),
      max([(e - s) / 1e3 for s, e, _ in draws] or [0]), mean([g for _, _, g in draws]), 100.0 * sum(e - s for s, e, _ in draws) / (t1 - half)))
print("blob copy launches: %d, mean %.1f us" % (len(copies), mean([(e - s) / 1e3 for s, e in copies])))
print("observation launches: %d, mean %.0f us; with a draw launch beside more than half of them: %d, mean %.0f us; with none: %d, mean %.0f us" % (
      len(ras), mean([(e - s) / 1e3 for s, e in ras]), len(w), mean(w), len(wo), mean(wo)))
print("step launches: %d, mean %.0f us; busy %.0f %% of the time" % (len(steps), mean([(e - s) / 1e3 for s, e in steps]), 100.0 * sum(e - s for s, e in steps) / (t1 - half)))
