#!/bin/bash
# round 3: does a better-converged adaptive setup (more refinement passes / relaxation) save an outer iteration?
OUT=gpurun_out/${1:-r03al}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
run() {  # name, python dict update
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
run base      '{}'
run sweeps2   '{"setup_sweeps": 2}'
run sweeps1   '{"setup_sweeps": 1}'
run maxit16   '{"setup_maxiter": 16}'
run maxit24   '{"setup_maxiter": 24}'
run refine0   '{"setup_refine": 0}'
run sw4       '{"setup_sweeps": 4}'
run mx40      '{"setup_maxiter": 40}'
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    try: d = json.load(open(f))
    except Exception: print(f, "unreadable"); continue
    c = d["config"]
    print("%-16s value %8.1f ms/step %7.2f iters %s setup %.2f" % (f.split("/")[-1], d["value"], d["ms_per_step"], c["outer_iterations_max"], c.get("setup_s") or 0))
PY
