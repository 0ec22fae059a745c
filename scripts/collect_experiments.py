"""The result lines of a round's GPU calls, one section per call, from what gpurun merged back into gpurun_out/<tag>/ -- the record that
DESIGN.md's "tag" references point at (profiles/r08_experiments.txt).  gpurun_out/ is scratch; this file is what is kept.

    python scripts/collect_experiments.py r08 > profiles/r08_experiments.txt
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("raster_", "step_", "tower_draw", "frame_setup")


def header_of(script):
    out = []
    if os.path.exists(script):
        for line in open(script):
            if line.startswith("#!"):
                continue
            if not line.startswith("#"):
                break
            out.append(line[1:].strip())
    return " ".join(out)


def bench_line(path):
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception:
        return None
    cfg = d.get("config", {})
    legs = {k[6:]: round(v / 1e6, 2) for k, v in d.items() if k.startswith("value_") and isinstance(v, (int, float))}
    s = "%-44s %6.2f M obs/s  %d steps  %s ticks/call" % (os.path.basename(path), d["value"] / 1e6, d["steps"], cfg.get("ticks_per_call", "?"))
    if d.get("roofline"):
        s += "  pass %.1f us/tick" % (d["roofline"]["avg_launch_ms"] * 1e3)
    if d.get("roofline_physics"):
        s += "  step %.1f us/tick" % (d["roofline_physics"]["avg_launch_ms"] * 1e3)
    if legs:
        s += "  " + " ".join("%s=%s" % kv for kv in sorted(legs.items()))
    return s


def stats_lines(path):
    out = []
    try:
        rows = list(csv.reader(l for l in open(path) if not l.startswith("#")))
    except Exception:
        return out
    for r in rows[1:]:
        if len(r) >= 7 and any(k in r[0] for k in KERNELS) and r[1].isdigit() and len(out) < 4:
            out.append("    %-90s calls %5s  avg %9.1f us" % (r[0][:90], r[1], float(r[3]) / 1e3))
        elif len(r) == 5 and any(k in r[0] for k in ("raster_fast_batch", "raster_union_batch", "raster_glist_batch")):   # counter rows: kernel, counter, launches, mean, sum
            out.append("    %-60s %-24s per launch %14.1f" % (r[0][:60], r[1], float(r[3])))
    return out


def main():
    prefix = sys.argv[1] if len(sys.argv) > 1 else "r08"
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", prefix + "*"))):
        tag = os.path.basename(d)
        if not os.path.isdir(d):
            continue
        script = os.path.join(ROOT, "scripts", "gpu_%s.sh" % tag)
        for alt in ("gpu_%s_final.sh" % tag[:-1], "gpu_%s_counters.sh" % tag[:-2]):   # (r12z: gpu_r12_final.sh, r12zc: gpu_r12_counters.sh)
            if not os.path.exists(script) and tag[-1:] in "zc" and os.path.exists(os.path.join(ROOT, "scripts", alt)):
                script = os.path.join(ROOT, "scripts", alt)
        if not os.path.exists(script):   # (one-off scripts move to scripts/archive/ with their round's closing pass)
            script = os.path.join(ROOT, "scripts", "archive", "gpu_%s.sh" % tag)
        print("== %s (%s)" % (tag, os.path.relpath(script, ROOT) if os.path.exists(script) else "no script kept"))
        h = header_of(script)
        if h:
            print("   " + h)
        for f in sorted(os.listdir(d)):
            p = os.path.join(d, f)
            if f.endswith(".json"):
                line = bench_line(p)
                if line:
                    print(line)
            elif f.endswith(".csv"):
                ls = stats_lines(p)
                if ls:
                    print("  " + f)
                    print("\n".join(ls))
            elif f.startswith("pytest") and f.endswith(".log"):
                tail = [l.strip() for l in open(p, errors="replace").read().strip().splitlines()[-3:] if "passed" in l or "failed" in l or "error" in l.lower()]
                if tail:
                    print("  %s: %s" % (f, tail[-1]))
            elif f.endswith("census.txt"):
                for l in open(p, errors="replace").read().strip().splitlines()[:3]:
                    print("  " + l[:400])
        print()


if __name__ == "__main__":
    main()
