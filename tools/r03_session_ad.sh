#!/bin/bash
# round 3: coarsening structures of the synthetic 1024^2 hierarchy (config 5 says "3-level MG")
OUT=gpurun_out/${1:-r03ad}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="timeout -k 10 300 python bench.py --workload synthetic --lattice 1024 --nb 64 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
run() {  # name, coarsening json, cycle json
  CFG=$(python - <<PY
import json
from deflatedmlmc_schwinger_amd import hierarchy as H
c = H.synthetic_solver_cfg(1024, 10, "device")
c["coarsening"] = json.loads('$2')
c["cycle"] = json.loads('$3')
c["eo_levels"] = list(range(len(c["coarsening"])))
print(json.dumps(c))
PY
)
  $B --cfg "$CFG" > $OUT/b_$1.json 2> $OUT/b_$1.err || { echo "$1 failed"; tail -3 $OUT/b_$1.err; }
}
run base5   '[[8,8],[2,8],[2,8],[2,8]]' '[[0,10,0],[0,8,2],[0,6,0],[0,10,0]]'
run l4_a    '[[8,8],[4,8],[2,8]]'       '[[0,10,0],[0,10,2],[0,14,0]]'
run l4_b    '[[8,8],[2,8],[4,8]]'       '[[0,10,0],[0,10,2],[0,14,0]]'
run l3_a    '[[8,8],[8,8]]'             '[[0,10,0],[0,14,0]]'
run l3_k    '[[8,8],[8,8]]'             '[[0,10,2],[0,14,0]]'
run l3_16   '[[8,8],[8,16]]'            '[[0,10,0],[0,14,0]]'
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable"); continue
    sb = d.get("step_breakdown_ms") or {}
    c = d["config"]
    print("%-14s value %8.1f ms/step %7.2f iters %s launches %s mvm %.2f coarsest %.2f levels %s setup %.2f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], c["outer_iterations_max"], sb.get("kernel_launches"),
        sb.get("mvm", 0), sb.get("coarsest", 0), (c.get("solver") or {}).get("levels"), c.get("setup_s") or 0))
PY
