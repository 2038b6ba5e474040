#!/usr/bin/env python3
"""Per-(kernel, grid size) table from three rocprofv3 runs of the SAME command (tools/profile_pmc.sh): kernel-trace
durations, a --pmc FETCH_SIZE pass and a --pmc WRITE_SIZE pass.  FETCH_SIZE / WRITE_SIZE are KiB counters of the
L2's fabric-side requests; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, HBM;
re-measured for this engine's read shapes in profiles/r02_pmc_calibration.txt).  Infinity-Cache hits are counted, so
below the cache's capacity the figure is a fabric-side rate.
usage: summarize_pmc.py OUTDIR [min_total_ms] -> OUTDIR/pmc_table.txt, OUTDIR/pmc_table.json"""
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("HIP_vector_type<double, 2u>", "cplx").replace("HIP_vector_type<float, 2u>", "cplxf")
    return name.split("(")[0].replace("void ", "").replace("swk::", "")[:64]


def newest(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def main(out, min_ms=0.2):
    trace = newest(os.path.join(out, "stats", "**", "*kernel_trace.csv"))
    dur = {}
    if trace:
        for r in csv.DictReader(open(trace)):
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            k = (short(r["Kernel_Name"]), grid)
            d = dur.setdefault(k, [0, 0.0])
            d[0] += 1
            d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3       # us
    pmc = {}
    for key, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = newest(os.path.join(out, key, "**", "*counter_collection.csv"))
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            a = pmc.setdefault(k, {}).setdefault(counter, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    rows = []
    for k, (n, us) in dur.items():
        p = pmc.get(k, {})
        f = p.get("FETCH_SIZE")
        w = p.get("WRITE_SIZE")
        fmb = 2.0 * f[1] / f[0] * 1024 / 1e6 if f else None
        wmb = w[1] / w[0] * 1024 / 1e6 if w else None
        rows.append({"kernel": k[0], "grid": k[1], "launches": n, "avg_us": us / n, "total_ms": us * 1e-3,
                     "fetch_MB_x2": fmb, "write_MB": wmb,
                     "fabric_TBs": ((fmb or 0.0) + (wmb or 0.0)) / (us / n) if (fmb is not None or wmb is not None) else None})
    rows.sort(key=lambda r: -r["total_ms"])
    tot = sum(r["total_ms"] for r in rows)
    lines = ["%-64s %9s %7s %9s %9s %6s %10s %9s %8s" % ("kernel", "grid", "calls", "avg_us", "total_ms", "%", "fetch_MBx2",
                                                         "write_MB", "TB/s")]
    fmt = lambda v, p: ("%" + p) % v if v is not None else "-"
    for r in rows:
        if r["total_ms"] < min_ms:
            continue
        lines.append("%-64s %9d %7d %9.1f %9.2f %6.1f %10s %9s %8s" % (
            r["kernel"], r["grid"], r["launches"], r["avg_us"], r["total_ms"], 100 * r["total_ms"] / tot,
            fmt(r["fetch_MB_x2"], ".1f"), fmt(r["write_MB"], ".1f"), fmt(r["fabric_TBs"], ".2f")))
    lines.append("total kernel time %.1f ms in %d launches" % (tot, sum(r["launches"] for r in rows)))
    text = "\n".join(lines)
    open(os.path.join(out, "pmc_table.txt"), "w").write(text + "\n")
    json.dump({"rows": [r for r in rows if r["total_ms"] >= min_ms],
               "note": "per-launch averages per (kernel, grid size); FETCH_SIZE doubled (gfx950), KiB counters; "
                       "fabric_TBs = (fetch x 2 + write) / average duration; Infinity-Cache hits are counted"},
              open(os.path.join(out, "pmc_table.json"), "w"), indent=1)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.2)
