#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats + PMC passes) of tools/profile_round.sh into the small
text/JSON summaries that are committed under profiles/."""
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("HIP_vector_type<double, 2u>", "cplx").replace("HIP_vector_type<float, 2u>", "cplxf")
    return name.split("(")[0].replace("void ", "")[:70]


def main(out, extra=""):
    lines = []
    newest = lambda fs: sorted(fs, key=os.path.getmtime)[-1:]      # gpurun merges runs into one tree
    stats = newest(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        lines.append("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 4 --warmup 1 "
                     "--no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs" + extra)
        lines.append("%-72s %7s %12s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
        for r in rows[:24]:
            lines.append("%-72s %7d %12.3f %10.2f %6.2f" % (
                short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
        lines.append("total kernel time %.1f ms" % (tot / 1e6))
    pmc = {}
    for key, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        files = newest(glob.glob(os.path.join(out, key, "**", "*counter_collection.csv"), recursive=True))
        if not files:
            continue
        # per kernel, only the launches of its LARGEST grid size: the profiled process also
        # runs the device-side setup, whose launches of the same kernels work on 64 columns
        rows = [r for r in csv.DictReader(open(files[0])) if r.get("Counter_Name") == counter]
        grids = {}
        for r in rows:
            g = grids.setdefault(short(r["Kernel_Name"]), {})
            g[r["Grid_Size"]] = g.get(r["Grid_Size"], 0) + 1
        main_grid = {k: max(g, key=lambda q: int(q)) for k, g in grids.items()}
        acc = {}
        for r in rows:
            k = short(r["Kernel_Name"])
            if r["Grid_Size"] != main_grid[k]:
                continue
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        pmc[counter] = acc
    summary = {}
    if pmc:
        lines.append("")
        lines.append("PMC passes (separate runs): per-launch averages over the launches of each kernel's largest")
        lines.append("grid size (the solve's; the device-side setup launches the same kernels on 64 columns); KiB counters;")
        lines.append("FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md, HBM; the")
        lines.append("factor 2 was re-measured for this engine's three read shapes, profiles/r02_pmc_calibration.txt)")
        names = sorted(set(list(pmc.get("FETCH_SIZE", {})) + list(pmc.get("WRITE_SIZE", {}))))
        lines.append("%-72s %12s %12s %12s" % ("kernel", "fetch_MB(x2)", "write_MB", "hbm_MB"))
        for k in names:
            f = pmc.get("FETCH_SIZE", {}).get(k)
            w = pmc.get("WRITE_SIZE", {}).get(k)
            fmb = 2.0 * f[1] / f[0] * 1024 / 1e6 if f else float("nan")
            wmb = w[1] / w[0] * 1024 / 1e6 if w else float("nan")
            lines.append("%-72s %12.2f %12.2f %12.2f" % (k, fmb, wmb, fmb + wmb))
            tags = {"k_stencil<0,": "k_stencil<0>", "k_stencil<1,": "k_stencil<1>",
                    "k_stencil<2,": "k_stencil<2>",
                    "k_schur_step<cplx, 2>": "k_schur_step",
                    "k_schur_step<cplx >": "k_schur_step",
                    "k_schur_step<cplx, 0>": "k_schur_step<0/1> (S x, b' - S x)",
                    "k_dense_mfma3_lds<4>": "k_bsr_mfma(dense coarsest)",
                    "k_dense_mfma3_lds<2>": "k_bsr_mfma(dense coarsest)",
                    "k_bsr_mfma3<0, 1, false, 8>": "k_bsr_mfma(dense coarsest)",
                    "k_bsr_mfma3<0, 2, false, 8>": "k_bsr_mfma(dense coarsest)",
                    "k_bsr_mfma3<3, 1, true": "k_bsr_mfma(level-1 operator)",
                    "k_bsr_mfma3<1, 1, true": "k_bsr_mfma(level-1 operator, residual form)",
                    "k_bsr_mfma<0, 4, false": "k_bsr_mfma(dense coarsest)",
                    "k_bsr_mfma<0, 2, false": "k_bsr_mfma(dense coarsest)",
                    "k_bsr_mfma<3, 4, true": "k_bsr_mfma(level-1 operator)",
                    "k_bsr_mfma<3, 4, false": "k_bsr_mfma(level-1 operator)",
                    "k_bsr_mfma<3, 2, true": "k_bsr_mfma(level-2 operator)",
                    "k_bsr_mfma<3, 2, false": "k_bsr_mfma(level-2 operator)"}
            for pat, tag in tags.items():
                if pat in k.replace("k_stencil<0, ", "k_stencil<0,").replace(
                        "k_stencil<1, ", "k_stencil<1,").replace("k_stencil<2, ", "k_stencil<2,"):
                    summary[tag] = {"hbm_bytes_per_launch": (fmb + wmb) * 1e6,
                                    "fetch_bytes_per_launch_corrected": fmb * 1e6,
                                    "write_bytes_per_launch": wmb * 1e6}
    # the even-odd smoother class of bench.py ("k_schur_step") in product form: launch-weighted mean over its
    # factor launches <3>, the fused last factors <4> and one residual <1> per smoothing pass (= per <4>)
    def _mb(counter, key, scale):
        v = pmc.get(counter, {}).get(key)
        return (scale * v[1] / v[0] * 1024 / 1e6, v[0]) if v else None
    k3, k4, k1 = ("swk::k_schur_step<cplx, %d>" % m for m in (3, 4, 1))
    if pmc and all(_mb("FETCH_SIZE", k, 2.0) and _mb("WRITE_SIZE", k, 1.0) for k in (k3, k4, k1)):
        n3, n4 = _mb("FETCH_SIZE", k3, 2.0)[1], _mb("FETCH_SIZE", k4, 2.0)[1]
        wts = {k3: n3, k4: n4, k1: n4}
        tot = float(sum(wts.values()))
        fmb = sum(_mb("FETCH_SIZE", k, 2.0)[0] * w for k, w in wts.items()) / tot
        wmb = sum(_mb("WRITE_SIZE", k, 1.0)[0] * w for k, w in wts.items()) / tot
        summary["k_schur_step"] = {"hbm_bytes_per_launch": (fmb + wmb) * 1e6,
                                   "fetch_bytes_per_launch_corrected": fmb * 1e6,
                                   "write_bytes_per_launch": wmb * 1e6,
                                   "composition": "product form: %d launches <3>, %d <4>, %d <1>" % (n3, n4, n4)}
        lines.append("%-72s %12.2f %12.2f %12.2f" % ("class k_schur_step (smoother, product form: <3>, <4>, <1>)",
                                                     fmb, wmb, fmb + wmb))
    # hierarchies whose first block level is small run it on the NT = 2 variant: bench.py calls the
    # first block level "level-1 operator" whatever its size
    l1, l2 = "k_bsr_mfma(level-1 operator)", "k_bsr_mfma(level-2 operator)"
    if l1 not in summary and l2 in summary:
        summary[l1] = summary.pop(l2)
    text = "\n".join(lines)
    open(os.path.join(out, "summary.txt"), "w").write(text + "\n")
    if summary:
        summary["note"] = ("per-launch averages over one bench step (one stream); FETCH_SIZE doubled per "
                           "MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B; factor re-measured "
                           "for 16-B, 8-B and segmented 8-B reads in profiles/r02_pmc_calibration.txt)")
        json.dump(summary, open(os.path.join(out, "kernel_pmc.json"), "w"), indent=1)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], " " + " ".join(sys.argv[2:]) if len(sys.argv) > 2 else "")
