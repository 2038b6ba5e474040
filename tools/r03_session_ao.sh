#!/bin/bash
# round 3: rocprofv3 kernel statistics of the secondary workloads (synthetic 1024^2, config 2 as written)
OUT=gpurun_out/${1:-r03ao}
mkdir -p $OUT
export OMP_NUM_THREADS=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s1024 -- python3 bench.py --workload synthetic --lattice 1024 --nb 128 --steps 2 --warmup 1 --no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs > $OUT/b1024.json 2> $OUT/b1024.err || { tail -5 $OUT/b1024.err; exit 1; }
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -- python3 bench.py --workload config2 --steps 2 --warmup 1 --no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs > $OUT/c2.json 2> $OUT/c2.err || { tail -5 $OUT/c2.err; exit 1; }
python3 - <<PY
import csv, glob, json, re
def short(n):
    n = n.replace("HIP_vector_type<double, 2u>", "cplx").replace("void ", "")
    return n.split("(")[0][:78]
for tag, cmd in (("s1024", "bench.py --workload synthetic --lattice 1024 --nb 128 --steps 2 --warmup 1"), ("c2", "bench.py --workload config2 --steps 2 --warmup 1")):
    f = glob.glob("$OUT/%s/**/*kernel_trace.csv" % tag, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "k_pack_i8" in r["Kernel_Name"]]
    a, b = marks[-2], marks[-1]
    sel = rows[a:b]
    span = (int(rows[b]["Start_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
    agg = {}
    for r in sel:
        k = short(r["Kernel_Name"])
        v = agg.setdefault(k, [0, 0])
        v[0] += 1; v[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in agg.values())
    out = ["rocprofv3 --kernel-trace -- python3 %s   (one timed batch of the run: the launches between two k_pack_i8)" % cmd,
           "batch span %.2f ms under the profiler, %d launches, kernel time %.2f ms" % (span, len(sel), tot / 1e6),
           "%-80s %6s %10s %10s %6s" % ("kernel", "calls", "avg_us", "total_ms", "%")]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        out.append("%-80s %6d %10.1f %10.3f %6.1f" % (k, v[0], v[1] / v[0] / 1e3, v[1] / 1e6, 100.0 * v[1] / tot))
    open("$OUT/%s_batch.txt" % tag, "w").write("\n".join(out) + "\n")
    print("\n".join(out)); print()
PY
