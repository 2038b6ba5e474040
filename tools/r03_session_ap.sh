#!/bin/bash
# round 3, final state: restart length of the Gram-matrix cycles once more (3 = default)
OUT=gpurun_out/${1:-r03ap}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
run() {
  CFG=$(python - <<PY
import json
from deflatedmlmc_schwinger_amd import hierarchy as H
c = dict(H.TUNED_SOLVER_CFG_128)
c.update($2)
print(json.dumps(c))
PY
)
  $B --cfg "$CFG" > $OUT/b_$1.json 2> $OUT/b_$1.err || { echo "$1 failed"; tail -3 $OUT/b_$1.err; }
}
run m3a '{"restart": 3}'
run m4a '{"restart": 4}'
run m3b '{"restart": 3}'
run m4b '{"restart": 4}'
run m4s '{"restart": 4, "cycle": [[0, 7, 0], [0, 10, 0]]}'
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    try: d = json.load(open(f))
    except Exception: print(f, "unreadable"); continue
    sb = d.get("step_breakdown_ms") or {}
    print("%-12s value %8.1f ms/step %7.2f iters %s launches %s dots %.2f axpy %.2f mvm %.2f" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"), sb.get("dots",0), sb.get("axpy",0), sb.get("mvm",0)))
PY
