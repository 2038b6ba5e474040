#!/bin/bash
# round 3: config 2 as written with the coarse level solved in even-odd form (ref_coarsest = "eo")
OUT=gpurun_out/${1:-r03aj}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 900 python -m pytest tests/test_gpu_golden.py -x -q -m gpu -k "config2_as_written" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
B2="timeout -k 10 300 python bench.py --workload config2 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
for c in eo dense; do
  SW_CONFIG2_COARSEST=$c $B2 > $OUT/c2_$c.json 2> $OUT/c2_$c.err || { tail -5 $OUT/c2_$c.err; exit 1; }
done
for nu in 48 80; do
  SW_CONFIG2_NU=$nu $B2 > $OUT/c2_eo_nu$nu.json 2> $OUT/c2_eo_nu$nu.err || { tail -5 $OUT/c2_eo_nu$nu.err; }
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/c2*.json")):
    try: d = json.load(open(f))
    except Exception: print(f, "unreadable"); continue
    sb = d.get("step_breakdown_ms") or {}
    print("%-18s value %8.1f ms/step %7.2f iters %s launches %s mvm %.2f coarsest %.2f setup %.1f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"),
        sb.get("mvm", 0), sb.get("coarsest", 0), d["config"].get("setup_s") or 0))
PY
