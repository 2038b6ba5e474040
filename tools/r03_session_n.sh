#!/bin/bash
# round 3: re-tune restart length / Schur steps with the Gram-form cycles (BLAS-1 at half its former cost)
OUT=gpurun_out/${1:-r03n}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
cfg() { python - "$@" <<PY
import json, sys
from deflatedmlmc_schwinger_amd import hierarchy
c = dict(hierarchy.TUNED_SOLVER_CFG_128)
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1)
    c[k] = json.loads(v)
print(json.dumps(c))
PY
}
for m in 2 3 4; do for nu in 7 8 9 10; do
  $B --cfg "$(cfg restart=$m "cycle=[[0,$nu,0],[0,10,0]]")" > $OUT/b_m${m}_nu${nu}.json 2> $OUT/b_m${m}_nu${nu}.err || { tail -3 $OUT/b_m${m}_nu${nu}.err; exit 1; }
done; done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    d = json.load(open(f))
    sb = d.get("step_breakdown_ms") or {}
    print("%-20s value %8.1f ms/step %7.2f iters %s launches %s dots %.2f axpy %.2f mvm %.2f coarsest %.2f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"),
        sb.get("dots", 0), sb.get("axpy", 0), sb.get("mvm", 0), sb.get("coarsest", 0)))
PY
