#!/bin/bash
# round 3, final state: cycle parameters and kernel switches on the synthetic 1024^2 lattice, 128-probe batches
OUT=gpurun_out/${1:-r03am}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="timeout -k 10 300 python bench.py --workload synthetic --lattice 1024 --nb 128 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
run() {  # name, cycle json, restart, engine opts
  CFG=$(python - <<PY
import json
from deflatedmlmc_schwinger_amd import hierarchy as H
c = H.synthetic_solver_cfg(1024, 10, "device")
c["cycle"] = json.loads('$2')
c["restart"] = $3
print(json.dumps(c))
PY
)
  $B --cfg "$CFG" --engine-opts "$4" > $OUT/b_$1.json 2> $OUT/b_$1.err || { echo "$1 failed"; tail -3 $OUT/b_$1.err; }
}
run l1_8     '[[0,10,0],[0,8,2],[0,8,0],[0,14,0]]' 3 ""
run l1_8l2_6 '[[0,10,0],[0,8,2],[0,6,0],[0,14,0]]' 3 ""
run l1_8l2_7 '[[0,10,0],[0,8,2],[0,7,0],[0,14,0]]' 3 ""
run l1_7     '[[0,10,0],[0,7,2],[0,8,0],[0,14,0]]' 3 ""
run l1_6     '[[0,10,0],[0,6,2],[0,8,0],[0,14,0]]' 3 ""
run l1_8l3_10 '[[0,10,0],[0,8,2],[0,8,0],[0,10,0]]' 3 ""
run l1_8l3_18 '[[0,10,0],[0,8,2],[0,8,0],[0,18,0]]' 3 ""
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    try: d = json.load(open(f))
    except Exception: print(f, "unreadable"); continue
    sb = d.get("step_breakdown_ms") or {}
    print("%-14s value %8.1f ms/step %7.2f iters %s launches %s mvm %.2f P %.2f R %.2f dots %.2f axpy %.2f coarsest %.2f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"),
        sb.get("mvm", 0), sb.get("P", 0), sb.get("R", 0), sb.get("dots", 0), sb.get("axpy", 0), sb.get("coarsest", 0)))
PY
