#!/bin/bash
# round 3: config 2 as written with the even-odd Schur polynomial on the reference hierarchy's lattice level
# (first pass, degrees 12-32 and restarts: gpurun_out r03s; this pass: degrees 40-64)
OUT=gpurun_out/${1:-r03s2}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="timeout -k 10 300 python bench.py --workload config2 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
for nu in 40 48 56 64; do
  SW_CONFIG2_SMOOTHER=eo SW_CONFIG2_NU=$nu $B > $OUT/b_eo$nu.json 2> $OUT/b_eo$nu.err || { tail -5 $OUT/b_eo$nu.err; exit 1; }
done
SW_CONFIG2_SMOOTHER=eo SW_CONFIG2_NU=48 SW_CONFIG2_RESTART=4 $B > $OUT/b_eo48_m4.json 2> $OUT/b_eo48_m4.err
SW_CONFIG2_SMOOTHER=eo SW_CONFIG2_NU=48 SW_CONFIG2_RESTART=3 $B > $OUT/b_eo48_m3.json 2> $OUT/b_eo48_m3.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    d = json.load(open(f))
    sb = d.get("step_breakdown_ms") or {}
    print("%-20s value %8.1f ms/step %7.2f iters %s launches %s dots %.2f axpy %.2f mvm %.2f coarsest %.2f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"),
        sb.get("dots", 0), sb.get("axpy", 0), sb.get("mvm", 0), sb.get("coarsest", 0)))
PY
