#!/bin/bash
# round 3: product form of the even-odd smoother (eo_product) against the step form: tests, then A/B of the bench
OUT=gpurun_out/${1:-r03w}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "product_form or time_skewed" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
B="timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
i=0
for lds in 1 0 1 0; do
  i=$((i+1))
  n=$OUT/b_prod${lds}_$i
  $B --engine-opts "eo_product=$lds" > $n.json 2> $n.err || { tail -5 $n.err; exit 1; }
done
B5="timeout -k 10 300 python bench.py --workload synthetic --lattice 1024 --nb 64 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
for lds in 1 0; do
  $B5 --engine-opts "eo_product=$lds" > $OUT/b1024_prod$lds.json 2> $OUT/b1024_prod$lds.err || { tail -5 $OUT/b1024_prod$lds.err; exit 1; }
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b*.json")):
    d = json.load(open(f))
    sb = d.get("step_breakdown_ms") or {}
    r = d.get("roofline") or {}
    print("%-28s value %8.1f ms/step %7.2f iters %s launches %s mvm %.2f | %s avg %.2f us frac %.3f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"),
        sb.get("mvm", 0), r.get("kernel"), 1e3 * (r.get("avg_launch_ms") or 0), r.get("frac") or 0))
PY
# config 2 as written (degree-64 even-odd smoother) in both forms
B2="timeout -k 10 300 python bench.py --workload config2 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
for prod in 1 0; do
  $B2 --engine-opts "eo_product=$prod" > $OUT/c2_prod$prod.json 2> $OUT/c2_prod$prod.err || { tail -5 $OUT/c2_prod$prod.err; exit 1; }
  python -c "
import json; d=json.load(open('$OUT/c2_prod$prod.json')); print('config2 eo_product=$prod value %.1f ms/step %.2f iters %s' % (d['value'], d['ms_per_step'], d['config']['outer_iterations_max']))"
done
