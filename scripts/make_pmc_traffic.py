#!/usr/bin/env python
"""profiles/<tag>_pmc_*.csv (rocprofv3 PMC passes summarised by scripts/rocpd_summary.py) -> profiles/pmc_traffic.json, the per-launch figures
bench.py reads back for `roofline.traffic` / `roofline.valu`:  python scripts/make_pmc_traffic.py r03a"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path):
    out = {}
    if not os.path.exists(path):
        return out
    rows = list(csv.reader(open(path)))
    i = next(k for k, r in enumerate(rows) if r[:2] == ["kernel", "counter"])
    for r in rows[i + 1:]:
        if len(r) >= 5:
            out.setdefault(r[0], {})[r[1]] = (int(r[2]), float(r[3]))
    return out


def pick(d, needle):
    for k, v in d.items():
        if needle in k:
            return k, v
    return None, {}


def main():
    tag = sys.argv[1]
    P = lambda n: os.path.join(ROOT, "profiles", f"{tag}_pmc_{n}.csv")
    wr, fe, sq, sq2 = counters(P("WRITE_SIZE")), counters(P("FETCH_SIZE")), counters(P("SQ")), counters(P("SQ2"))
    out = {"source": f"profiles/{tag}_pmc_WRITE_SIZE.csv + {tag}_pmc_FETCH_SIZE.csv + {tag}_pmc_SQ.csv + {tag}_pmc_SQ2.csv (rocprofv3 --pmc, separate passes, mean "
                     "per dispatch; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; unit KB)",
           "config": {"envs_per_gpu": 1024, "agents_per_env": 1, "obs": [128, 128]}, "kernels": {}}
    for key, needle in (("raster", "raster_fast_kernel"), ("step", "step_kernel")):
        kn, w = pick(wr, needle)
        _, f = pick(fe, needle)
        _, s1 = pick(sq, needle)
        _, s2 = pick(sq2, needle)
        if not w or not f:
            continue
        wb, fb = w["WRITE_SIZE"][1] * 1024.0, f["FETCH_SIZE"][1] * 1024.0 * 2.0
        e = {"kernel": kn, "write_bytes": wb, "fetch_bytes": fb, "traffic_bytes_per_launch": wb + fb, "launches": w["WRITE_SIZE"][0]}
        if s1:
            e["valu"] = {"valu_insts_per_launch": s1["SQ_INSTS_VALU"][1], "salu_insts_per_launch": s2.get("SQ_INSTS_SALU", (0, None))[1],
                         "lds_insts_per_launch": s1.get("SQ_INSTS_LDS", (0, None))[1], "wave_cycles": s1.get("SQ_WAVE_CYCLES", (0, None))[1],
                         "wait_inst_any": s1.get("SQ_WAIT_INST_ANY", (0, None))[1], "wait_any": s2.get("SQ_WAIT_ANY", (0, None))[1],
                         "active_inst_any": s2.get("SQ_ACTIVE_INST_ANY", (0, None))[1], "source": f"profiles/{tag}_pmc_SQ.csv, {tag}_pmc_SQ2.csv"}
        out["kernels"][key] = e
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
