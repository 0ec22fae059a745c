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
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 8   # ticks per launch of the batched kernels (bench.py's default): their counters are divided by it
    sys.path.insert(0, ROOT)
    from bench import kernel_sources_sha16
    P = lambda n: os.path.join(ROOT, "profiles", f"{tag}_pmc_{n}.csv")
    wr, fe, sq, sq2, sq3 = counters(P("WRITE_SIZE")), counters(P("FETCH_SIZE")), counters(P("SQ")), counters(P("SQ2")), counters(P("SQ3"))
    out = {"source": f"profiles/{tag}_pmc_WRITE_SIZE.csv + {tag}_pmc_FETCH_SIZE.csv + {tag}_pmc_SQ.csv + {tag}_pmc_SQ2.csv (rocprofv3 --pmc, separate passes, mean "
                     "per dispatch; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; unit KB)",
           "source_sha16": kernel_sources_sha16(),   # bench.py prints these figures only while the kernel sources are the ones the passes ran on
           "config": {"envs_per_gpu": 1024, "agents_per_env": 1, "obs": [128, 128]}, "kernels": {}}
    # the batched kernels (k ticks per launch; a step launch holds at most 8: a call of 16 is two of them) when the passes ran bench.py's default call size,
    # else the per-tick kernels; everything PER TICK
    for key, needles in (("raster", (("raster_fast_batch_kernel", ticks), ("raster_fast_kernel", 1))), ("step", (("step_ticks_pipe_kernel", min(ticks, 8)), ("step_ticks_kernel", min(ticks, 8)), ("step_kernel", 1)))):
        for needle, div in needles:
            kn, w = pick(wr, needle)
            if w:
                break
        _, f = pick(fe, needle)
        _, s1 = pick(sq, needle)
        _, s2 = pick(sq2, needle)
        _, s3 = pick(sq3, needle)
        if not w or not f:
            continue
        wb, fb = w["WRITE_SIZE"][1] * 1024.0 / div, f["FETCH_SIZE"][1] * 1024.0 * 2.0 / div
        e = {"kernel": kn, "ticks_per_launch": div, "write_bytes": wb, "fetch_bytes": fb, "traffic_bytes_per_launch": wb + fb, "launches": w["WRITE_SIZE"][0]}
        if s1:
            per = lambda d, k: (d[k][1] / div) if k in d else None
            e["valu"] = {"valu_insts_per_launch": per(s1, "SQ_INSTS_VALU"), "salu_insts_per_launch": per(s2, "SQ_INSTS_SALU"),
                         "lds_insts_per_launch": per(s1, "SQ_INSTS_LDS"), "wave_cycles": per(s1, "SQ_WAVE_CYCLES"),
                         "wait_inst_any": per(s1, "SQ_WAIT_INST_ANY"), "wait_any": per(s2, "SQ_WAIT_ANY"),
                         "active_inst_any": per(s2, "SQ_ACTIVE_INST_ANY"),
                         # quad-cycles (4 clocks) the SIMDs' vector ALUs were busy, summed over the chip's 1024 SIMDs: rocprofv3's derived VALUBusy is
                         # 100 x this x 4 / 1024 / the launch's clocks
                         "active_inst_valu_quadcycles": per(s1, "SQ_ACTIVE_INST_VALU"), "source": f"profiles/{tag}_pmc_SQ.csv, {tag}_pmc_SQ2.csv"}
        if s1 and "SQ_LDS_BANK_CONFLICT" in s1:
            # north_star's "LDS ... on the raster tile": LDS instructions per tick, the quad-cycles the LDS pipe was busy with them (SQ3 pass, when it ran) and
            # the quad-cycles it spent replaying bank conflicts; conflict_frac = conflict / busy ("hit rate" = 1 - conflict_frac: an LDS access has no miss,
            # only a replay)
            per = lambda d, k: (d[k][1] / div) if d and k in d else None
            busy = per(s3, "SQ_ACTIVE_INST_LDS")
            conf = per(s1, "SQ_LDS_BANK_CONFLICT")
            e["lds"] = {"insts_per_launch": per(s1, "SQ_INSTS_LDS"), "bank_conflict_quadcycles": conf, "active_inst_lds_quadcycles": busy,
                        "idx_active_quadcycles": per(s3, "SQ_LDS_IDX_ACTIVE"), "wait_inst_lds": per(s2, "SQ_WAIT_INST_LDS"),
                        "conflict_frac": (conf / busy) if busy else None, "conflict_per_inst": conf / per(s1, "SQ_INSTS_LDS") if per(s1, "SQ_INSTS_LDS") else None,
                        "source": f"profiles/{tag}_pmc_SQ.csv, {tag}_pmc_SQ2.csv, {tag}_pmc_SQ3.csv"}
        out["kernels"][key] = e   # ("per launch" in the field names: per TICK, the launch of the per-tick kernels)
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
