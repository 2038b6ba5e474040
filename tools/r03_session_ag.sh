#!/bin/bash
# round 3: k_schur_step with a capped grid whose workgroups walk the sites (schur_grid) against one workgroup per
# (the schur_grid option was removed after this session: profiles/r03_ab_sessions.txt, r03ag)
# four sites
OUT=gpurun_out/${1:-r03ag}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "grid_cap or product_form or time_skewed" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
B="timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
for g in 2048 0 1024 4096 2048 0 3072; do
  n=$OUT/b_grid${g}_$RANDOM
  $B --engine-opts "schur_grid=$g" > $n.json 2> $n.err || { tail -5 $n.err; exit 1; }
done
B5="timeout -k 10 300 python bench.py --workload synthetic --lattice 1024 --nb 64 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
for g in 2048 0 1024; do
  $B5 --engine-opts "schur_grid=$g" > $OUT/b1024_grid$g.json 2> $OUT/b1024_grid$g.err || { tail -5 $OUT/b1024_grid$g.err; exit 1; }
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
