#!/bin/bash
# round 3, first GPU session: the whole -m gpu suite, then A/B bench lines for the round's engine changes.
# usage (GPU box): bash tools/r03_session_a.sh <outdir under gpurun_out>
OUT=gpurun_out/${1:-r03a}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $OUT/gputests.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a $OUT/gputests.log
tail -n 40 $OUT/gputests.log
if [ $rc -gt 1 ]; then exit $rc; fi
B="timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil"
DIRECT='{"coarsening": [[8, 8], [2, 8]], "cycle": [[0, 9, 0], [0, 10, 0]], "smoother": "richardson", "eo_levels": [0, 1], "restart": 3, "setup": "device", "setup_sweeps": 3, "setup_tol": 0.1, "setup_maxiter": 32, "setup_refine": 1, "direct_levels": [1]}'
$B > $OUT/b_default.json 2> $OUT/b_default.err && \
$B --engine-opts fused_reduce=0 > $OUT/b_unfused.json 2> $OUT/b_unfused.err && \
$B --engine-opts stop_factor=0.1 > $OUT/b_strict.json 2> $OUT/b_strict.err && \
$B --streams 1 > $OUT/b_s1.json 2> $OUT/b_s1.err && \
$B --streams 2 > $OUT/b_s2.json 2> $OUT/b_s2.err && \
$B --cfg "$DIRECT" --streams 1 > $OUT/b_direct_s1.json 2> $OUT/b_direct_s1.err && \
$B --cfg "$DIRECT" --streams 2 > $OUT/b_direct_s2.json 2> $OUT/b_direct_s2.err && \
$B --cfg "$DIRECT" --streams 3 > $OUT/b_direct_s3.json 2> $OUT/b_direct_s3.err
echo "bench rc=$?"
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    try:
        d = json.load(open(f))
        sb = d.get("step_breakdown_ms") or {}
        print("%-28s value %8.0f  ms/step %7.2f  iters %s  launches %s  dots %.2f axpy %.2f mvm %.2f coarsest %.2f other %.2f"
              % (f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"],
                 sb.get("kernel_launches"), sb.get("dots", 0), sb.get("axpy", 0), sb.get("mvm", 0),
                 sb.get("coarsest", 0), sb.get("other", 0)))
    except Exception as e:
        print(f, "unreadable:", e)
PY
