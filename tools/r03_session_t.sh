#!/bin/bash
# round 3: where the time goes on the synthetic 1024^2 lattice (config 5): per-kernel statistics of one bench run
OUT=gpurun_out/${1:-r03t}
mkdir -p $OUT
export OMP_NUM_THREADS=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --workload synthetic --lattice 1024 --nb 64 --steps 2 --warmup 1 --no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs > $OUT/bench1024.json 2> $OUT/bench1024.err || { tail -5 $OUT/bench1024.err; exit 1; }
python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print("%-90s calls %7s avg %10.1f us  total %8.2f ms  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, 100*float(r["TotalDurationNs"])/tot))
d = json.load(open("$OUT/bench1024.json"))
print(d["value"], d["ms_per_step"], d.get("step_breakdown_ms"))
PY
