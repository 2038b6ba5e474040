#!/bin/bash
# round 3: smoother polynomial fitted on what the coarse correction leaves (smoother_target = complement)
OUT=gpurun_out/${1:-r03p}
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
for nu in 6 7 8; do
  $B --cfg "$(cfg 'smoother_target="complement"' "cycle=[[0,$nu,0],[0,10,0]]")" > $OUT/b_compl_nu${nu}.json 2> $OUT/b_compl_nu${nu}.err || { tail -3 $OUT/b_compl_nu${nu}.err; exit 1; }
done
$B > $OUT/b_all_nu8.json 2> $OUT/b_all_nu8.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    d = json.load(open(f))
    sb = d.get("step_breakdown_ms") or {}
    print("%-20s value %8.1f ms/step %7.2f iters %s launches %s mvm %.2f coarsest %.2f setup %.2f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"),
        sb.get("mvm", 0), sb.get("coarsest", 0), d["config"]["setup_s"]))
PY
